"""GPU parity tests shaped like BASELINE.json configs[3] and configs[4] at sizes the oracle and a
single test run can afford: the same code paths (M = 32 scan template, nlist = 16384 coarse,
nprobe = 64; d = 768 inner product with dsub = 12 codebooks, range-filter bitmaps, inserts and
updates interleaved with searches), checked against the CPU oracle on the same inputs."""
import numpy as np
import pytest

from gamma_amd import api, synth
from tests import lloyd as train
from oracle import binding as B
from tests.parity import compare_search_exact, compare_exact

pytestmark = pytest.mark.gpu
WIDE = dict(min_score=-3e38, max_score=3e38)


def _oracle_from_device(g, d, nlist, M, metric, cc, pq, base, bucket=4000):
    o = B.OracleIVFPQ(d, nlist, M, 8, metric, bucket_init_size=bucket)
    o.set_trained(cc, pq, g.ivfpq_table())
    for l in range(nlist):
        ids, codes = g.get_list(l)
        if len(ids):
            o.add_keys(l, ids, codes)
    o.set_raw(base)
    return o


def test_c4_shape_m32_nlist16384_nprobe64():
    import torch
    N, d, nlist, M, P = 1500000, 128, 16384, 32, 64
    base = synth.sift_like(N, d=d, seed=1234)
    cc, pq = train.train_ivfpq(base[:nlist * 40], nlist, M, niter=6, pq_niter=10, seed=3,
                               device="cuda" if torch.cuda.is_available() else "cpu")
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=200)
        g.ivfpq_set_trained(cc, pq, None)
        for i0 in range(0, N, 500000):
            g.add(base[i0:i0 + 500000], i0)
        g.raw_init(d)
        g.raw_append(base)
        assert sum(g.list_size(l) for l in range(nlist)) == N
        q = synth.sift_like(512, d=d, seed=4321)
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=100, has_rank=True, coarse_mode=1,
                              min_score=0.0, max_score=1e30)
        D, I = g.ivfpq_search(q, 10, args)
        assert (np.diff(D, axis=1) >= 0).all() and (I >= 0).all() and (I < N).all()
        ex = ((base[I[:50].ravel()] - np.repeat(q[:50], 10, axis=0)) ** 2).sum(1).reshape(50, 10)
        assert np.array_equal(ex.astype(np.float32), D[:50])
        Df, If = g.flat_search(q[:100], 10, api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30))
        rec = np.mean([len(set(I[i].tolist()) & set(If[i].tolist())) / 10.0 for i in range(100)])
        assert rec > 0.85, rec
        # sampled bit parity against the oracle holding the same lists
        o = _oracle_from_device(g, d, nlist, M, B.METRIC_L2, cc, pq, base)
        qs = q                       # all 512: 4 probes per scan workgroup, threshold pre-filter on
        for has_rank in (True, False):
            ctx = B.make_ctx(min_score=0.0, max_score=1e30)
            Do, Io, st = o.search(qs, 10, P, recall_num=100, has_rank=has_rank, metric=B.METRIC_L2, ctx=ctx,
                                  coarse_mode=1, want_stages=True)
            a2 = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=100, has_rank=has_rank,
                                coarse_mode=1, min_score=0.0, max_score=1e30)
            Dg, Ig = g.ivfpq_search(qs, 10, a2)
            sg = g.last_stages(len(qs), P, 100)
            assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
            compare_exact(st["coarse_dis"], st["coarse_idx"], sg["coarse_dis"], sg["coarse_idx"])   # tie-aware
            compare_search_exact(Do, Io, st, Dg, Ig, sg)
    finally:
        g.close()


def test_c5_shape_d768_ip_filters_and_realtime_inserts():
    import torch
    N, d, nlist, M, P = 60000, 768, 1024, 64, 32
    rng = np.random.default_rng(11)
    # unit-normalised Gaussian mixture (embedding-shaped)
    centres = rng.standard_normal((256, d)).astype(np.float32)
    lab = rng.integers(0, 256, size=N + 3000)
    allv = centres[lab] + 0.6 * rng.standard_normal((N + 3000, d)).astype(np.float32)
    allv /= np.linalg.norm(allv, axis=1, keepdims=True)
    allv = np.ascontiguousarray(allv, dtype=np.float32)
    base, pool = allv[:N], allv[N:]
    q = pool[-64:]
    scalar = rng.integers(0, 1000000, size=N + 3000)          # the int field range filters select on
    cc, pq = train.train_ivfpq(base[:nlist * 40], nlist, M, niter=6, pq_niter=8, seed=5,
                               device="cuda" if torch.cuda.is_available() else "cpu")
    g = api.GammaHip(0)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_IP, bucket_init_size=100)
    o.set_trained(cc, pq, None)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_IP, bucket_init_size=100)
        g.ivfpq_set_trained(cc, pq, None)
        assert g.ivfpq_table().tobytes() == o.table().tobytes()
        g.raw_init(d)
        # engine-sized Add batches (<= 1000 vectors, vector/vector_manager.cc:305-349), device encode
        B.lib().go_set_assign_mode(1)
        n_added = 0

        def add(x):
            nonlocal n_added
            g.raw_append(x)
            g.add(x, n_added)
            assert o.add(x)
            n_added += len(x)

        for i0 in range(0, N, 1000):
            add(base[i0:i0 + 1000])
        for l in range(0, nlist, 37):
            ids, codes = o.get_list(l)
            gi, gc = g.get_list(l)
            assert np.array_equal(ids, gi) and np.array_equal(codes, gc)

        def check(nvec, del_bm=None):
            o.set_raw(allv[:nvec])
            for sel in (None, 0.01, 0.10, 0.50):
                rf_o = rf_g = None
                if sel is not None:
                    docs = np.nonzero(scalar[:nvec] < int(sel * 1000000))[0]
                    rf_o = [B.make_range_filter(docs)]
                    rf_g = [api.make_range_filter(docs)]
                for has_rank in (True, False):
                    ctx = B.make_ctx(docids_bitmap=del_bm, range_filters=rf_o, **WIDE)
                    Do, Io, st = o.search(q, 10, P, recall_num=100, has_rank=has_rank, metric=B.METRIC_IP,
                                          ctx=ctx, coarse_mode=1, want_stages=True)
                    a = api.SearchArgs(metric=api.METRIC_IP, nprobe=P, recall_num=100, has_rank=has_rank,
                                       coarse_mode=1, range_filters=rf_g, **WIDE)
                    Dg, Ig = g.ivfpq_search(q, 10, a)
                    sg = g.last_stages(len(q), P, 100)
                    compare_search_exact(Do, Io, st, Dg, Ig, sg)

        check(N)
        # realtime inserts between searches, then deletes + updates
        for b in range(2):
            add(pool[b * 1000:(b + 1) * 1000])
            check(n_added)
        dead = rng.choice(n_added, size=500, replace=False)
        bm = np.zeros((n_added >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        g.bitmap_upload(bm, n_added)
        g.delete(dead)
        o.set_docids_bitmap(bm)
        o.delete(dead)
        B.lib().go_set_assign_mode(0)
        for vid in rng.choice(n_added, size=40, replace=False):
            newv = pool[2000 + int(vid) % 900]
            lno, code = g.encode(newv[None, :])
            g.update(int(lno[0]), int(vid), code[0])
            g.raw_update(int(vid), newv)
            allv[vid] = newv
            o.update(int(vid), newv)
        check(n_added, del_bm=bm)
        # a batch large enough for the scan's threshold pre-filter (generic M = 64 kernel, inner product)
        qb = np.ascontiguousarray(allv[rng.choice(n_added, size=520, replace=False)] * np.float32(0.999))
        o.set_raw(allv[:n_added])
        for has_rank in (True, False):
            ctx = B.make_ctx(docids_bitmap=bm, **WIDE)
            Do, Io, st = o.search(qb, 10, P, recall_num=100, has_rank=has_rank, metric=B.METRIC_IP, ctx=ctx,
                                  coarse_mode=1, want_stages=True)
            a = api.SearchArgs(metric=api.METRIC_IP, nprobe=P, recall_num=100, has_rank=has_rank, coarse_mode=1,
                               **WIDE)
            Dg, Ig = g.ivfpq_search(qb, 10, a)
            sg = g.last_stages(len(qb), P, 100)
            compare_search_exact(Do, Io, st, Dg, Ig, sg)
    finally:
        B.lib().go_set_assign_mode(0)
        g.close()


def test_c1_flat_l2_10k_128_k10_through_the_plugin_boundary():
    """BASELINE.json configs[0] at its exact shape: Flat L2, 10 000 x 128 float32, k = 10, driven the way the
    engine drives a model (reflector -> HIPFLAT -> Init / Add in engine-sized batches / Parse + Search with a
    GammaSearchCondition, gamma_amd/host/harness_c_api.cc), 1000 queries as in SURVEY 8d -- against the CPU
    restatement of GammaFLATIndex::Search.  Also the engine's default score window (tests/test.h:584-585)."""
    from gamma_amd import plugin
    from oracle import binding as B
    N, d, k, nq = 10000, 128, 10, 1000
    base = synth.sift_like(N, d=d, seed=1234)
    q = synth.sift_like(nq, d=d, seed=4321)
    m = plugin.PluginModel("HIPFLAT", d, '{"metric_type": "L2"}')
    m.store(base)
    for i0 in range(0, N, 1000):              # AddRTVecsToIndex: batches of <= 1000 (vector_manager.cc:305-349)
        assert m.add(base[i0:i0 + 1000])
    Df, If = B.flat_search(base, q, k, B.METRIC_L2, B.make_ctx())
    Dg, Ig = m.search(q, k, '{"metric_type": "L2"}')
    compare_exact(Df, If, Dg, Ig)
    assert (Ig >= 0).all() and (np.diff(Dg, axis=1) >= 0).all()
    Df, If = B.flat_search(base, q, k, B.METRIC_L2, B.make_ctx(min_score=0.0, max_score=10000.0))
    Dg, Ig = m.search(q, k, '{"metric_type": "L2"}', min_score=0.0, max_score=10000.0)
    compare_exact(Df, If, Dg, Ig)
    # one query at a time, as tests/test.h issues them
    for i in range(0, 40):
        D1, I1 = m.search(q[i:i + 1], k, "")
        compare_exact(Df[i:i + 1] * 0 + B.flat_search(base, q[i:i + 1], k, B.METRIC_L2, B.make_ctx())[0],
                     B.flat_search(base, q[i:i + 1], k, B.METRIC_L2, B.make_ctx())[1], D1, I1)
    m.close()


def test_default_path_is_the_librarys_default_blas_path_at_the_c3_shape():
    """tests/golden/ivfpq_blas_c3shape.npz: a C3-shaped index (200 k x 128, 1024 lists, M 16) trained, filled and searched
    by the COMPILED library at its default distance_compute_blas_threshold -- coarse quantizer through MKL sgemm_
    (faiss:utils/distances.cpp:215-296,303-305).  The device's DEFAULT path (coarse_mode -1, exact ties on, Add through the
    device encode with faiss's assign rule) must return exactly that for a 2048-query batch: lists, coarse distances bit
    for bit, probe order, final labels at every rank, distances bit for bit.  Mismatch bound: zero queries."""
    from tests.test_oracle_golden import check_blas_c3shape_lists, load_blas_c3shape
    z, base, q = load_blas_c3shape()
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    nprobe, R, k = int(z["nprobe"]), int(z["R"]), int(z["k"])
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 1000)
        g.ivfpq_set_trained(z["cc"], z["pq"], None)
        g.raw_init(d)
        for i0 in range(0, len(base), 50000):
            g.raw_append(base[i0:i0 + 50000])
            g.add(base[i0:i0 + 50000], i0)
        check_blas_c3shape_lists(z, g.get_list)
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=nprobe, recall_num=R, has_rank=True, coarse_mode=-1, **WIDE)
        for nq in (len(q), 700, 40):           # matrix-free coarse path from 4096 queries on is covered by the bench leg
            D, I = g.ivfpq_search(q[:nq], k, args)
            sg = g.last_stages(nq, nprobe, R)
            assert sg["coarse_dis"].tobytes() == z["coarse_dis_blas"][:nq].tobytes()
            assert np.array_equal(sg["coarse_idx"], z["coarse_idx_blas"][:nq].astype(np.int64))
            compare_exact(z["D_blas"][:nq], z["I_blas"][:nq].astype(np.int64), D, I)
        # below 20 queries the library itself takes the exact form
        D, I = g.ivfpq_search(q[:19], k, args)
        compare_exact(z["D_exact"][:19], z["I_exact"][:19].astype(np.int64), D, I)
        # a large batch (the matrix-free coarse kernels, the bounded scan): the fixture's queries tiled
        reps = 3
        D, I = g.ivfpq_search(np.tile(q, (reps, 1)), k, args)
        compare_exact(np.tile(z["D_blas"], (reps, 1)), np.tile(z["I_blas"].astype(np.int64), (reps, 1)), D, I)
        assert g.ties_not_honoured() == 0
    finally:
        g.close()


@pytest.mark.parametrize("d,nlist", [(768, 256), (512, 200), (448, 130), (392, 64)])
def test_gemm_form_k_split_matches_the_oracle(d, nlist):
    """384 < d <= 768: the compiled sgemm_ sums K in two blocks (kernels.h gemm_k_split, oracle go_gemm_k_split, pinned
    against the library in tests/test_oracle_vs_ref.py).  The three device kernels that compute the GEMM form for long
    rows -- k_l2_gemmform_big<true> (batches from 256 queries) and k_l2_gemmform_mfma (smaller ones; d = 392 splits inside a
    K slab) -- against the oracle's mode 1, bit for bit, through the coarse entry point."""
    import torch
    rng = np.random.default_rng(d)
    cc = (rng.standard_normal((nlist, d)) * 2).astype(np.float32)
    M = min(64, d // 8)
    pq = (rng.standard_normal((M, 256, d // M)) * 0.1).astype(np.float32)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 100)
        g.ivfpq_set_trained(cc, pq, None)
        P = 16
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, coarse_mode=1, **WIDE)
        for nq in (24, 300, 1000):
            x = rng.standard_normal((nq, d)).astype(np.float32)
            Do, Io = B.knn_L2sqr(x, cc, P, mode=1)
            dev = torch.device("cuda", 0)
            dx = torch.from_numpy(x).to(dev)
            cd = torch.empty((nq, P), dtype=torch.float32, device=dev)
            pr = torch.empty((nq, P), dtype=torch.int32, device=dev)
            g.ivfpq_coarse_device(dx.data_ptr(), nq, args, cd.data_ptr(), pr.data_ptr())
            g.synchronize()
            assert cd.cpu().numpy().tobytes() == Do.tobytes(), (d, nq)
            assert np.array_equal(pr.cpu().numpy().astype(np.int64), Io), (d, nq)
    finally:
        g.close()


def test_filter_pass_with_several_consumer_groups():
    """Long lists: the scan's filter pass cuts the probes behind the producer's into several consumer groups
    (gamma_hip_search.cpp, cf_span).  The shapes of this suite have short lists, so the split is forced here through
    GAMMA_HIP_SCAN_CF_CODES (read once per process: a child process) on the bounded-scan parity tests and the C3 headline
    test -- strict comparisons, as in the parent."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAMMA_HIP_SCAN_CF_CODES="1500")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_more.py", "tests/test_gpu_ties.py",
                        "-k", "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch_search"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_list_major_byte_table_pass_on_short_lists():
    """The list-major consumer pass over byte tables (csrc/q8scan.hip: one list x 8 queries per tile, candidates recomputed
    exactly per query) is the default from 800 codes per list on -- the C4 / C5 list-length regimes above run it.  Here it is
    forced on the short-list suites (GAMMA_HIP_Q8_MINLEN=0, read once per process: a child process): the bounded-scan
    parity tests, the C3 headline test and the large-batch fuzz -- strict comparisons, as in the parent."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAMMA_HIP_Q8_MINLEN="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_more.py", "tests/test_gpu_ties.py",
                        "tests/test_gpu_fuzz.py", "-k",
                        "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_filter_pass_on_the_fp32_table():
    """GAMMA_HIP_NO_C8=1: the consumers' filter pass gathers from the fp32 table (rounds 3-4) instead of its byte image, and
    the first probe group goes back to eight lists.  The default is the byte image (scan.hip, ScanBound::c8) -- the bounded-scan
    parity tests, the C3 headline test, the tie suites and the large-batch fuzz run it in the parent; here the same tests in a
    child process with the variable set, strict comparisons."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAMMA_HIP_NO_C8="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_more.py", "tests/test_gpu_ties.py",
                        "tests/test_gpu_fuzz.py", "-k",
                        "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch or ivfpq_exact_ties or cut_ties"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_producer_on_the_filter_pass_arithmetic():
    """GAMMA_HIP_PROD_CF=1 (scan.hip, ScanBound::prod_cf; off by default -- it measured slower): the producer workgroup scores
    its probes with the query's table + the per-code sums, bounds from those approximate values plus their error margin,
    recomputes only its candidates exactly, and the first group's slab segment is re-scored for every query whose slab is read
    (unfiltered selection, tie replay).  The bounded-scan parity tests, the C3 headline test, the tie suites and the large-batch
    fuzz in a child process with the variable set: strict comparisons, as in the parent."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAMMA_HIP_PROD_CF="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_more.py", "tests/test_gpu_ties.py",
                        "tests/test_gpu_fuzz.py", "-k",
                        "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch or ivfpq_exact_ties or cut_ties"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_producer_on_the_byte_image():
    """GAMMA_HIP_PROD_C8 (scan.hip, ScanBound::prod_c8): the producer workgroup scores its probe group on the byte image of the
    query's table as the consumers do -- lower estimates in LDS, their recall_num-th smallest plus the image's proven error width
    bounds the recall_num-th best exact value, only the codes under it get the reference's arithmetic -- and does not write the
    group's slab segment: that is scored by the repair launch for every query whose slab is read (unfiltered selection, tie
    replay).  The bounded-scan parity tests, the C3 headline test, the tie suites and the large-batch fuzz in a child process
    with the variable set to the other value than the parent's default: strict comparisons, as in the parent."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    other = "0" if os.environ.get("GAMMA_HIP_PROD_C8", PROD_C8_DEFAULT) != "0" else "1"
    env = dict(os.environ, GAMMA_HIP_PROD_C8=other)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_more.py", "tests/test_gpu_ties.py",
                        "tests/test_gpu_fuzz.py", "-k",
                        "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch or ivfpq_exact_ties or cut_ties"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


PROD_C8_DEFAULT = "0"   # the library's default (csrc/gamma_hip_search.cpp, prod_c8_on)


def test_bounded_scan_backs_off_where_the_bound_is_loose():
    """The scan's pre-filter bounds a query's recall_num-th best from its NEAREST probe group.  Inner-product data whose
    best candidates sit in lists far from the query in L2 (centroids s_l * u with scales 0.5 .. 2: the quantizer probes
    the lists at s ~ 1 first, the scores grow with s): the bound is loose, the survivor slices of the far groups overflow,
    every query falls through to the unfiltered selection -- after the filtered scan.  The handle counts that
    (gamma_hip_scan_bound_stats) and turns the pre-filter off for such calls (full-size C5: 41 -> 24 ms per 4096 queries).
    Results are the oracle's before and after, and with the feedback off."""
    rng = np.random.default_rng(77)
    d, nlist, M, N, P, R, k = 32, 64, 8, 64000, 64, 100, 10
    u = rng.standard_normal(d).astype(np.float32)
    u /= np.linalg.norm(u)
    scales = np.linspace(0.5, 2.0, nlist).astype(np.float32)
    cc = (scales[:, None] * u[None, :] + 0.01 * rng.standard_normal((nlist, d))).astype(np.float32)
    lab = rng.integers(0, nlist, size=N)
    base = (scales[lab, None] * (u[None, :] + 0.05 * rng.standard_normal((N, d)))).astype(np.float32)
    _, pq = api.train_ivfpq(base[:20000], nlist, M)
    q1 = (u[None, :] + 0.05 * rng.standard_normal((64, d))).astype(np.float32)
    q = np.tile(q1, (64, 1))                       # 4096 queries per call
    B.lib().go_set_assign_mode(1)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_IP)
    o.set_trained(cc, pq, None)
    assert o.add(base)
    B.lib().go_set_assign_mode(0)
    o.set_raw(base)
    D1, I1 = o.search(q1, k, P, recall_num=R, has_rank=True, metric=B.METRIC_IP, ctx=B.make_ctx(**WIDE), coarse_mode=1)
    De, Ie = np.tile(D1, (64, 1)), np.tile(I1, (64, 1))
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_IP)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base)
        g.add(base, 0)
        args = api.SearchArgs(metric=api.METRIC_IP, nprobe=P, recall_num=R, has_rank=True, **WIDE)
        g.set_scan_bound_feedback(False)
        for _ in range(8):
            D, I = g.ivfpq_search(q, k, args)
            compare_exact(De, Ie, D, I)
        st0 = g.scan_bound_stats()
        assert st0["queries"] == 8 * len(q) and st0["fell_through"] > st0["queries"] // 2, st0   # the data does its job
        assert st0["backoffs"] == 0
        g.set_scan_bound_feedback(True)
        for _ in range(12):
            D, I = g.ivfpq_search(q, k, args)
            compare_exact(De, Ie, D, I)
        st1 = g.scan_bound_stats()
        assert st1["backoffs"] >= 1, st1
        assert st1["queries"] - st0["queries"] < 12 * len(q)      # some of the 12 calls ran without the pre-filter
        # another kind of call starts over: a narrow bound (few probes of the nearest lists) stays filtered
        a2 = api.SearchArgs(metric=api.METRIC_IP, nprobe=16, recall_num=R, has_rank=True, **WIDE)
        D2o, I2o = o.search(q1, k, 16, recall_num=R, has_rank=True, metric=B.METRIC_IP, ctx=B.make_ctx(**WIDE), coarse_mode=1)
        for _ in range(3):
            D, I = g.ivfpq_search(q, k, a2)
            compare_exact(np.tile(D2o, (64, 1)), np.tile(I2o, (64, 1)), D, I)
    finally:
        g.close()


# ---- the list-length regimes of the full-size 8-GPU configurations -----------------------------------
# Full-size C4 has 6100 codes per list, C5 2440: the regime in which the scan's filter pass is OFF (mean list length above
# GAMMA_HIP_SCAN_CF_MAXLEN), consumer groups are split, the bounded-scan feedback acts, and MT = 32 / 64 run at batch size.
# The tests above keep nlist at the configurations' values and therefore have 60-90 codes per list.  Here nlist is cut
# instead, so that the oracle affords lists as long as the real ones; >= 4096 queries per call (the oracle's 512 tiled).


def _tile_stages(st, reps):
    return {k: np.tile(v, (reps, 1)) for k, v in st.items()}


def _check_regime(g, o, q1, reps, metric, P, R, k, has_rank, ctx_kw=None, args_kw=None, win=WIDE):
    bm = B.METRIC_L2 if metric == api.METRIC_L2 else B.METRIC_IP
    D1, I1, st1 = o.search(q1, k, P, recall_num=R, has_rank=has_rank, metric=bm, ctx=B.make_ctx(**win, **(ctx_kw or {})),
                           coarse_mode=1, want_stages=True)
    q = np.tile(q1, (reps, 1))
    Dg, Ig = g.ivfpq_search(q, k, api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=has_rank, coarse_mode=1,
                                                 **win, **(args_kw or {})))
    sg = g.last_stages(len(q), P, max(R, k))
    compare_search_exact(np.tile(D1, (reps, 1)), np.tile(I1, (reps, 1)), _tile_stages(st1, reps), Dg, Ig, sg)
    assert g.ties_not_honoured() == 0
    return D1, I1


@pytest.mark.parametrize("M,nlist,N", [(16, 256, 230000), (32, 192, 180000)])
def test_lists_around_the_switch_between_the_two_consumer_passes(M, nlist, N):
    """~900 codes per list: the region where the list-major byte-table pass takes over from the query-major one (800 codes per
    list, csrc/gamma_hip_search.cpp: the lists of the large-batch fuzz stop at ~780) -- its short-list kernel (k_q8_filter_sl) at
    batch size, M 16 and 32, has_rank both, deletes + a 10 % range filter; then the SAME calls with the switch moved out of the way
    (GAMMA_HIP_Q8_MINLEN is read once per process, so the other side runs in the fuzz and the short-list suites), all against the
    oracle: probe order, recall-stage sets, labels at every rank."""
    d, P, k = 128, 32, 10
    base = synth.sift_like(N, d=d, seed=99)
    cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=2000)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base)
        g.add(base, 0)
        sizes = np.array([g.list_size(l) for l in range(nlist)])
        assert sizes.sum() == N and 800 < sizes.mean() < 1000
        o = _oracle_from_device(g, d, nlist, M, B.METRIC_L2, cc, pq, base, bucket=2000)
        q1 = synth.sift_like(512, d=d, seed=4321)
        for R in (100, 200):
            for has_rank in (True, False):
                _check_regime(g, o, q1, 8, api.METRIC_L2, P, R, k, has_rank)
        rng = np.random.default_rng(5)
        dead = rng.choice(N, N // 20, replace=False)
        bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        g.bitmap_upload(bm, N)
        g.delete(dead)
        o.set_docids_bitmap(bm)
        o.delete(dead)
        docs = np.nonzero(rng.random(N) < 0.10)[0]
        _check_regime(g, o, q1, 8, api.METRIC_L2, P, 200, k, True, ctx_kw=dict(docids_bitmap=bm))
        _check_regime(g, o, q1, 8, api.METRIC_L2, P, 200, k, True,
                      ctx_kw=dict(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)]),
                      args_kw=dict(range_filters=[api.make_range_filter(docs)]))
    finally:
        g.close()


def test_c4_list_length_regime_m32_5900_codes_per_list():
    """C4's regime: 1.5 M x 128, M 32, nprobe 64 over 256 lists (5 900 codes per list; full-size C4: 6 100), 4096 queries
    per call, recall_num 100 / 150 / 300 (the configuration's recall bar sits at 150; 300 is beyond the bounded scan's old
    gate), has_rank both, then a delete bitmap and a 10 % range filter -- every call against the oracle holding the
    same lists: probe order, recall-stage (distance, id) sets, labels at every rank (gamma_index_ivfpq.cc:701-890)."""
    N, d, nlist, M, P, k = 1500000, 128, 256, 32, 64, 10
    base = synth.sift_like(N, d=d, seed=1234)
    cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=8000)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        for i0 in range(0, N, 500000):
            g.raw_append(base[i0:i0 + 500000])
            g.add(base[i0:i0 + 500000], i0)
        sizes = np.array([g.list_size(l) for l in range(nlist)])
        assert sizes.sum() == N and sizes.mean() > 5000
        o = _oracle_from_device(g, d, nlist, M, B.METRIC_L2, cc, pq, base, bucket=8000)
        q1 = synth.sift_like(512, d=d, seed=4321)
        for R in (100, 150, 300):
            for has_rank in (True, False):
                _check_regime(g, o, q1, 8, api.METRIC_L2, P, R, k, has_rank)
        # the engine's default score window on a has_rank call
        _check_regime(g, o, q1, 8, api.METRIC_L2, P, 150, k, True, win=dict(min_score=0.0, max_score=1e30))
        rng = np.random.default_rng(5)
        dead = rng.choice(N, N // 20, replace=False)
        bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        g.bitmap_upload(bm, N)
        g.delete(dead)
        o.set_docids_bitmap(bm)
        o.delete(dead)
        docs = np.nonzero(rng.random(N) < 0.10)[0]
        for R, has_rank in ((100, True), (300, True), (150, False)):
            _check_regime(g, o, q1, 8, api.METRIC_L2, P, R, k, has_rank, ctx_kw=dict(docids_bitmap=bm))
            _check_regime(g, o, q1, 8, api.METRIC_L2, P, R, k, has_rank,
                          ctx_kw=dict(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)]),
                          args_kw=dict(range_filters=[api.make_range_filter(docs)]))
        # fewer queries than a full batch (other probe-group sizes, the small-batch chain's long-list units)
        for nq1 in (1, 7, 64, 300):
            _check_regime(g, o, q1[:nq1], 1, api.METRIC_L2, P, 150, k, True, ctx_kw=dict(docids_bitmap=bm))
    finally:
        g.close()


def test_c5_list_length_regime_d768_ip_m64_2300_codes_per_list():
    """C5's regime: 300 k x 768 inner product, M 64 (dsub 12), nprobe 64 over 128 lists (2 300 codes per list; full-size C5:
    2 440), 4096 queries per call, recall_num 100 and 1000 (the short-list at which full-size C5 reaches recall@10 0.95),
    the bounded scan's feedback on and off, then range filters of 1 % / 10 % / 50 % -- against the oracle, strict."""
    N, d, nlist, M, P, k = 300000, 768, 128, 64, 64, 10
    base = synth.embedding_like(N, d=d, seed=1234)
    cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_IP, bucket_init_size=3000)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        for i0 in range(0, N, 100000):
            g.raw_append(base[i0:i0 + 100000])
            g.add(base[i0:i0 + 100000], i0)
        sizes = np.array([g.list_size(l) for l in range(nlist)])
        assert sizes.sum() == N and sizes.mean() > 2000
        o = _oracle_from_device(g, d, nlist, M, B.METRIC_IP, cc, pq, base, bucket=3000)
        q1 = synth.embedding_like(512, d=d, seed=4321)
        for fb in (True, False):
            g.set_scan_bound_feedback(fb)
            for R in (100, 1000):
                for has_rank in (True, False):
                    for _ in range(3 if fb else 1):      # the feedback acts from the second call of a kind on
                        _check_regime(g, o, q1, 8, api.METRIC_IP, P, R, k, has_rank)
        g.set_scan_bound_feedback(True)
        rng = np.random.default_rng(9)
        scalar = rng.integers(0, 1000000, size=N)
        for sel in (0.01, 0.10, 0.50):
            docs = np.nonzero(scalar < int(sel * 1000000))[0]
            for R in (100, 1000):
                _check_regime(g, o, q1, 8, api.METRIC_IP, P, R, k, True,
                              ctx_kw=dict(range_filters=[B.make_range_filter(docs)]),
                              args_kw=dict(range_filters=[api.make_range_filter(docs)]))
        for nq1 in (1, 16, 200):
            _check_regime(g, o, q1[:nq1], 1, api.METRIC_IP, P, 1000, k, True)
    finally:
        g.close()
