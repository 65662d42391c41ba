"""GPU parity tests, part 2: filters, flat, edge cases, golden vectors from real faiss, device
realtime lists, device encode, and full-size (1M) properties.  Everything goes through the
C ABI (gamma_amd.api -> libgamma_hip.so)."""
import os

import numpy as np
import pytest

from gamma_amd import api, synth
from oracle import binding as B
from tests import fixtures
from tests.parity import compare_search_exact, compare_exact

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
WIDE = dict(min_score=-3e38, max_score=3e38)


def run_both(case, g, q, k, nprobe, R, metric, has_rank, coarse_mode=0, ctx_kw=None, range_docs=None,
             del_bitmap=None, b_not_in=False):
    ctx_kw = ctx_kw or WIDE
    rf_o = rf_g = None
    if range_docs is not None:
        rf_o = [B.make_range_filter(r, b_not_in=b_not_in) for r in range_docs]
        rf_g = [api.make_range_filter(r, b_not_in=b_not_in) for r in range_docs]
    ctx = B.make_ctx(docids_bitmap=del_bitmap, range_filters=rf_o, **ctx_kw)
    D, I, st = case["oracle"].search(q, k, nprobe, recall_num=R, has_rank=has_rank, metric=metric,
                                     ctx=ctx, coarse_mode=coarse_mode, want_stages=True)
    args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank,
                          coarse_mode=coarse_mode, range_filters=rf_g, **ctx_kw)
    Dg, Ig = g.ivfpq_search(q, k, args)
    return (D, I, st), (Dg, Ig)


# --------------------------------------------------------------------------- filters / windows
@pytest.fixture(scope="module")
def case():
    return fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)


@pytest.fixture(scope="module")
def hip(case):
    g = fixtures.load_hip(case)
    yield g
    g.close()


@pytest.mark.parametrize("has_rank", [True, False])
def test_delete_bitmap_and_range_filters(case, hip, has_rank):
    rng = np.random.default_rng(1)
    N = case["N"]
    deleted = rng.choice(N, size=N // 5, replace=False)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, deleted >> 3, (1 << (deleted & 7)).astype(np.uint8))
    hip.bitmap_upload(bm, N)
    try:
        r1 = rng.choice(N, size=N // 2, replace=False)           # ~50 % selectivity
        r2 = np.arange(1000, 15000)                               # a contiguous docid range
        for ranges, not_in in (([r1], False), ([r1, r2], False), ([r2], True), ([], False)):
            (D, I, _), (Dg, Ig) = run_both(case, hip, case["q"], 10, 8, 100, B.METRIC_L2, has_rank,
                                           range_docs=ranges, del_bitmap=bm, b_not_in=not_in)
            compare_exact(D, I, Dg, Ig)
            live = Ig[Ig >= 0]
            assert not np.isin(live, deleted).any()
            if ranges == []:
                assert (Ig == -1).all()   # MultiRangeQueryResults::Has on an empty set is false
    finally:
        hip.bitmap_upload(np.zeros(1, dtype=np.uint8), 0)


@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
def test_score_window(case, hip, metric):
    # default GammaSearchCondition window: min = FLT_MIN (tiny), max = FLT_MAX
    (D, I, _), (Dg, Ig) = run_both(case, hip, case["q"], 10, 8, 100, metric, True,
                                   ctx_kw=dict(min_score=None, max_score=None))
    compare_exact(D, I, Dg, Ig)
    lo, hi = (20000.0, 60000.0) if metric == B.METRIC_L2 else (1e5, 4e5)
    for has_rank in (True, False):
        (D, I, _), (Dg, Ig) = run_both(case, hip, case["q"], 10, 8, 100, metric, has_rank,
                                       ctx_kw=dict(min_score=lo, max_score=hi))
        compare_exact(D, I, Dg, Ig)
        ok = Ig >= 0
        assert (Dg[ok] >= lo).all() and (Dg[ok] <= hi).all()


def test_edge_shapes(case, hip):
    q = case["q"]
    # nq = 1 (exact coarse rule), k = 1, recall_num < k is raised to k, nprobe = nlist
    for nq, k, nprobe, R in ((1, 1, 1, 1), (1, 10, 64, 5), (3, 200, 2, 50), (64, 10, 64, 300)):
        (D, I, _), (Dg, Ig) = run_both(case, hip, q[:nq], k, nprobe, R, B.METRIC_L2, True, coarse_mode=-1)
        compare_exact(D, I, Dg, Ig)
    # k <= 0 leaves the outputs alone and succeeds (gamma_index_ivfpq.cc:753-756)
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=4)
    Dg, Ig = hip.ivfpq_search(q[:2], 0, args)
    assert Dg.shape == (2, 0)
    # bad nprobe is rejected loudly
    with pytest.raises(api.GammaHipError):
        hip.ivfpq_search(q[:2], 5, api.SearchArgs(metric=api.METRIC_L2, nprobe=65))


def test_not_trained_and_missing_raw_fail_loudly():
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(16, 8, 4)
        with pytest.raises(api.GammaHipError):
            g.ivfpq_search(np.zeros((1, 16), np.float32), 3, api.SearchArgs(nprobe=2))
        with pytest.raises(api.GammaHipError):
            g.flat_search(np.zeros((1, 16), np.float32), 3, api.SearchArgs())
        with pytest.raises(api.GammaHipError):
            g2 = api.GammaHip(0)
            g2.ivfpq_init(16, 8, 4, nbits=4)     # only nbits == 8 on device
    finally:
        g.close()


# --------------------------------------------------------------------------- other shapes
@pytest.mark.parametrize("cfg", [
    dict(d=128, nlist=32, M=16, N=12000),          # dsub 8, 16-byte codes (uint4 path)
    dict(d=128, nlist=32, M=32, N=12000),          # dsub 4, 32-byte codes
    dict(d=96, nlist=16, M=8, N=8000),             # dsub 12
    dict(d=20, nlist=16, M=4, N=6000),             # dsub 5 (generic), d % 8 != 0
    dict(d=64, nlist=16, M=4, N=6000),             # dsub 16 (generic AVX order)
    dict(d=24, nlist=300, M=12, N=2000),           # dsub 2, many empty lists
    dict(d=128, nlist=8, M=64, N=12000),           # dsub 2, 64-byte codes: two codes per thread, lists of ~1500
])
@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
def test_other_shapes(cfg, metric):
    case = fixtures.trained_case(nq=24, metric=B.METRIC_L2, **cfg)
    g = fixtures.load_hip(case)
    try:
        assert g.ivfpq_table().tobytes() == case["oracle"].table().tobytes()
        for has_rank in (True, False):
            for cm in (0, 1):
                (D, I, st), (Dg, Ig) = run_both(case, g, case["q"], 7, min(6, cfg["nlist"]), 40, metric,
                                                has_rank, coarse_mode=cm)
                sg = g.last_stages(len(case["q"]), min(6, cfg["nlist"]), 40)
                assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
                compare_search_exact(D, I, st, Dg, Ig, sg)
    finally:
        g.close()


@pytest.mark.parametrize("R,k", [(64, 63), (65, 63), (65, 64), (130, 129), (200, 10), (200, 63), (200, 64), (257, 1), (1000, 63),
                                 (1024, 40), (1024, 200)])
@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
def test_rerank_reads_only_its_first_k_plus_one(case, hip, R, k, metric):
    """k_rerank_topk (csrc/rerank.hip, round 6): up to k + 1 = 64 the recall_num exact distances are not rank-sorted in full --
    64-item runs sorted by waves, the runs' first k + 1 items ranked against the other runs -- beyond it the full rank sort.
    recall_num on both sides of the run boundaries (64, 65, 130, 257, 1000, 1024), k + 1 on both sides of 64; results and the
    recall-stage tables identical to the oracle's at every rank, exact ties on."""
    (D, I, st), (Dg, Ig) = run_both(case, hip, case["q"], k, 24, R, metric, True, coarse_mode=1)
    sg = hip.last_stages(len(case["q"]), 24, R)
    compare_search_exact(D, I, st, Dg, Ig, sg)


# --------------------------------------------------------------------------- flat
@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
@pytest.mark.parametrize("d", [128, 32, 20])
def test_flat_parity(metric, d):
    N, nq, k = 70000, 16, 100            # > one 65536-row chunk: exercises the chunk merge
    base = synth.sift_like(N, d=d, seed=21)
    q = synth.sift_like(nq, d=d, seed=22)
    rng = np.random.default_rng(2)
    deleted = rng.choice(N, size=N // 10, replace=False)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, deleted >> 3, (1 << (deleted & 7)).astype(np.uint8))
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(base[:30000])
        g.raw_append(base[30000:])
        g.bitmap_upload(bm, N)
        keep = rng.choice(N, size=N // 3, replace=False)
        for ranges in (None, [keep]):
            rf_o = [B.make_range_filter(r) for r in ranges] if ranges else None
            rf_g = [api.make_range_filter(r) for r in ranges] if ranges else None
            ctx = B.make_ctx(docids_bitmap=bm, range_filters=rf_o, **WIDE)
            D, I = B.flat_search(base, q, k, metric, ctx)
            # calls of up to 64 queries take the whole store as one chunk and the small-batch chains' selection
            # (gamma_hip_search.cpp, flat_search_device_locked); without it: row chunks + merge.  3: the two-level
            # selection with three slices only (slices longer than the registers hold)
            res = []
            for mode in (0, 1, 3):
                g.set_small_path(mode)
                Dg, Ig = g.flat_search(q, k, api.SearchArgs(metric=metric, range_filters=rf_g, **WIDE))
                compare_exact(D, I, Dg, Ig)
                res.append((Dg, Ig))
            for Dg, Ig in res[1:]:
                assert res[0][0].tobytes() == Dg.tobytes()
    finally:
        g.close()


def test_flat_small_calls_are_the_chunked_search():
    """Flat search of 1 .. 64 queries: one row chunk + block selection against the chunked path, byte for byte -- duplicate
    rows (equal distances: ids in row order), deleted rows, a score window, k from 1 to more than the valid rows."""
    d, N = 24, 40000
    rng = np.random.default_rng(5)
    base = synth.sift_like(N, d=d, seed=31)
    base[1000:3000] = base[5000:7000]          # 2000 exact duplicates
    base[20000:20050] = base[0]
    q = np.concatenate([synth.sift_like(60, d=d, seed=32), base[[0, 5000, 1000, 39999]]])
    deleted = rng.choice(N, size=N // 5, replace=False)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, deleted >> 3, (1 << (deleted & 7)).astype(np.uint8))
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(base)
        for step in range(2):
            if step == 1:
                g.bitmap_upload(bm, N)
            for metric in (api.METRIC_L2, api.METRIC_IP):
                for nq, k in ((1, 1), (1, 1000), (3, 10), (64, 100), (17, 33)):
                    qq = q[-nq:]
                    g.set_small_path(0)
                    Dw, _ = g.flat_search(qq, k, api.SearchArgs(metric=metric, **WIDE))
                    fin = Dw[np.abs(Dw) < 1e37]
                    wins = [WIDE, dict(min_score=float(np.quantile(fin, 0.3)), max_score=float(np.quantile(fin, 0.9)))]
                    for kw in wins:
                        args = api.SearchArgs(metric=metric, **kw)
                        g.set_small_path(0)
                        D0, I0 = g.flat_search(qq, k, args)
                        for mode in (1, 2):
                            g.set_small_path(mode)
                            D1, I1 = g.flat_search(qq, k, args)
                            assert D0.tobytes() == D1.tobytes() and np.array_equal(I0, I1), (step, metric, nq, k, mode)
        # every valid row asked for and more (k > what is left after the deletes would need k > 1024: a small store)
        g2 = api.GammaHip(0)
        g2.raw_init(d)
        g2.raw_append(base[:300])
        for mode in (0, 1):
            g2.set_small_path(mode)
            r = g2.flat_search(q[:2], 400, api.SearchArgs(metric=api.METRIC_L2, **WIDE))
            if mode == 0:
                r0 = r
            else:
                assert r0[0].tobytes() == r[0].tobytes() and np.array_equal(r0[1], r[1])
        g2.close()
    finally:
        g.close()


@pytest.mark.parametrize("order", ["random", "improving", "ties"])
def test_flat_running_bound(order):
    """Beyond the first 65536 rows the flat search keeps per-query candidate lists under a running bound
    (row passes that double the rows seen).  random: the normal case; improving: rows sorted so that every
    pass beats the bound of the rows before it -- the lists overflow and the call is redone without a bound;
    ties: few distinct vectors, the k-th distance is shared by thousands of rows (membership by row id)."""
    N, d, nq = 300000, 32, 24
    base = synth.sift_like(N, d=d, seed=31)
    q = synth.sift_like(nq, d=d, seed=32)
    if order == "improving":
        dist0 = ((base - q[0]) ** 2).sum(1)
        base = np.ascontiguousarray(base[np.argsort(-dist0, kind="stable")])
    elif order == "ties":
        base = np.ascontiguousarray(base[np.random.default_rng(5).integers(0, 40, size=N)])
    rng = np.random.default_rng(3)
    deleted = rng.choice(N, size=N // 10, replace=False)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, deleted >> 3, (1 << (deleted & 7)).astype(np.uint8))
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(base)
        g.bitmap_upload(bm, N)
        for metric, k, win in ((B.METRIC_L2, 100, WIDE), (B.METRIC_IP, 10, WIDE), (B.METRIC_L2, 256, WIDE),
                               (B.METRIC_L2, 300, WIDE), (B.METRIC_L2, 100, dict(min_score=30000.0, max_score=1e30))):
            ctx = B.make_ctx(docids_bitmap=bm, **win)
            D, I = B.flat_search(base, q, k, metric, ctx)
            Dg, Ig = g.flat_search(q, k, api.SearchArgs(metric=metric, **win))
            compare_exact(D, I, Dg, Ig)      # "ties" too: membership and order inside the ties are the reference heap's
    finally:
        g.close()


@pytest.mark.parametrize("d,kind", [(128, "near_duplicates"), (64, "wide_range"), (96, "sift"), (32, "near_duplicates"), (128, "unit"),
                                    # long rows (round 5, k_flat_filter_big): C5's d = 768 and the variant's range, margins 2^-12 / 2^-11
                                    (768, "near_duplicates"), (768, "unit"), (256, "wide_range"), (512, "sift"), (1024, "unit"), (160, "unit"),
                                    (1536, "unit"), (1536, "near_duplicates"), (1280, "wide_range")])
def test_flat_matrix_filter_never_drops_a_neighbour(d, kind):
    """flat_mfma.hip: from 64 queries on the passes behind the first row chunk run a bf16 hi / lo filter on the matrix pipe
    (three products, a proven error margin) and only the survivors get the reference's exact arithmetic.  The filter must
    be a SUPERSET test: results identical to the oracle's exhaustive exact search, labels strictly -- on data built to sit
    on the margin: rows that are tiny perturbations of a few prototypes (thousands of distances within 1e-4 of the k-th),
    columns spanning six orders of magnitude, unit-norm embeddings; L2 and inner product; deletes, a range filter and a
    score window; batch sizes with padded tiles."""
    rng = np.random.default_rng(d)
    N = 90000 if d <= 128 else 40000
    if kind == "near_duplicates":
        proto = (rng.standard_normal((50, d)) * 40).astype(np.float32)
        base = (proto[rng.integers(0, 50, N)] + rng.standard_normal((N, d)).astype(np.float32) * np.float32(2e-3)).astype(np.float32)
        q = (proto[rng.integers(0, 50, 200)] + rng.standard_normal((200, d)).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
    elif kind == "wide_range":
        scale = (10.0 ** rng.uniform(-3, 3, d)).astype(np.float32)
        base = (rng.standard_normal((N, d)).astype(np.float32) * scale).astype(np.float32)
        q = (rng.standard_normal((200, d)).astype(np.float32) * scale).astype(np.float32)
    elif kind == "unit":
        base = rng.standard_normal((N, d)).astype(np.float32)
        base /= np.linalg.norm(base, axis=1, keepdims=True)
        q = rng.standard_normal((200, d)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    else:
        base = synth.sift_like(N, d=d, seed=7)
        q = synth.sift_like(200, d=d, seed=8)
    dead = rng.choice(N, N // 11, replace=False)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
    docs = rng.choice(N, 3 * N // 4, replace=False)
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(base)
        for step in range(2):
            ctx_kw, kw_f = {}, {}
            if step == 1:
                g.bitmap_upload(bm, N)
                ctx_kw = dict(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)])
                kw_f = dict(range_filters=[api.make_range_filter(docs)])
            for metric in (B.METRIC_L2, B.METRIC_IP):
                for nq, k in ((64, 10), (200, 100), (130, 1)):
                    wins = [WIDE]
                    if step == 1 and k == 10:
                        Dw, _ = B.flat_search(base, q[:nq], k, metric, B.make_ctx(**WIDE, **ctx_kw))
                        wins.append(dict(min_score=float(np.quantile(Dw, 0.2)), max_score=float(np.quantile(Dw, 0.9))))
                    for win in wins:
                        D, I = B.flat_search(base, q[:nq], k, metric, B.make_ctx(**win, **ctx_kw))
                        Dg, Ig = g.flat_search(q[:nq], k, api.SearchArgs(metric=metric, **win, **kw_f))
                        compare_exact(D, I, Dg, Ig)
    finally:
        g.close()


@pytest.mark.parametrize("k", [1, 10, 64])
@pytest.mark.parametrize("layout", ["random", "one_lane", "ties"])
def test_small_k_selection_paths(k, layout):
    """k <= 64 goes through the wave-per-row selection (select.hip): multi-chunk rows, the
    exact-extraction path (all near neighbours at positions = 5 mod 64, so the lane-minimum
    bound is useless) and rows that are one big tie."""
    d, N, nq = 16, 9000, 12
    rng = np.random.default_rng(7)
    base = synth.sift_like(N, d=d, seed=31)
    q = synth.sift_like(nq, d=d, seed=32)
    if layout == "one_lane":
        base = base + 400.0                       # everything far away ...
        near = np.arange(5, N, 64)
        base[near] = q[rng.integers(0, nq, size=len(near))] + rng.integers(0, 3, size=(len(near), d))
        base = base.astype(np.float32)
    elif layout == "ties":
        base[:] = base[0]                         # every distance equal: order = position
        base[100:140] = base[1]
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(base)
        for metric in (B.METRIC_L2, B.METRIC_IP):
            D, I = B.flat_search(base, q, k, metric, B.make_ctx(**WIDE))
            Dg, Ig = g.flat_search(q, k, api.SearchArgs(metric=metric, **WIDE))
            compare_exact(D, I, Dg, Ig)
            if layout == "ties":
                assert Dg.tobytes() == D.tobytes()
    finally:
        g.close()


def test_flat_small_and_empty():
    g = api.GammaHip(0)
    try:
        g.raw_init(16)
        q = synth.sift_like(3, d=16, seed=1)
        Dg, Ig = g.flat_search(q, 5, api.SearchArgs(metric=api.METRIC_L2, **WIDE))
        assert (Ig == -1).all() and (Dg == np.finfo(np.float32).max).all()
        base = synth.sift_like(3, d=16, seed=2)
        g.raw_append(base)
        D, I = B.flat_search(base, q, 5, B.METRIC_L2, B.make_ctx(**WIDE))
        Dg, Ig = g.flat_search(q, 5, api.SearchArgs(metric=api.METRIC_L2, **WIDE))
        compare_exact(D, I, Dg, Ig)      # k > N: -1 / FLT_MAX padding
    finally:
        g.close()


# --------------------------------------------------------------------------- golden (real faiss)
@pytest.mark.parametrize("name", ["ivfpq_l2_d32", "ivfpq_l2_d64", "ivfpq_ip_d48"])
def test_golden_vectors_from_real_faiss(name):
    z = np.load(os.path.join(G, name + ".npz"))
    d, nlist, M, N = int(z["d"]), int(z["nlist"]), int(z["M"]), int(z["N"])
    nprobe, R = int(z["nprobe"]), int(z["R"])
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, int(z["metric"]))
        g.ivfpq_set_trained(z["cc"], z["pq"], None)
        assert g.ivfpq_table().tobytes() == z["table"].tobytes()   # faiss precompute_table
        sizes = z["list_sizes"]
        nz = np.nonzero(sizes)[0]
        g.add_keys_batch(nz, sizes[nz], z["list_ids"], z["list_codes"])
        for m, tag in ((api.METRIC_L2, "l2"), (api.METRIC_IP, "ip")):
            args = api.SearchArgs(metric=m, nprobe=nprobe, recall_num=R, has_rank=False, coarse_mode=0, **WIDE)
            Dg, Ig = g.ivfpq_search(z["q"], 5, args)
            sg = g.last_stages(len(z["q"]), nprobe, R)
            assert sg["coarse_dis"].tobytes() == z["coarse_dis"].tobytes()
            assert np.array_equal(sg["coarse_idx"], z["coarse_idx"])
            # the stage table holds the reference's top-R as a SET (its order inside ties is the scan's; only a query whose
            # result a tie can change is replayed) ...
            from tests.parity import _stage_rows
            a_d, a_i = _stage_rows(z["rdis_" + tag], z["rids_" + tag])
            b_d, b_i = _stage_rows(sg["recall_dis"], sg["recall_ids"])
            assert np.array_equal(a_d, b_d) and np.array_equal(a_i, b_i)
            compare_exact(z["rdis_" + tag][:, :5], z["rids_" + tag][:, :5], Dg, Ig)
            # ... and asked for all R results, the call returns faiss's heap_reorder order rank by rank
            DR, IR = g.ivfpq_search(z["q"], R, args)
            compare_exact(z["rdis_" + tag], z["rids_" + tag], DR, IR)
    finally:
        g.close()


def test_golden_realtime_replay_on_device():
    """The reference's real RTInvertIndex script (AddKeys / Update / Delete / CompactIfNeed)
    replayed on the HBM-resident lists: same contents, same capacities."""
    z = np.load(os.path.join(G, "realtime.npz"))
    nlist, cs = int(z["nlist"]), int(z["cs"])
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(cs, nlist, cs, 8, api.METRIC_L2, int(z["binit"]), int(z["bmax"]))
        g.bitmap_upload(np.zeros(int(z["nbits"]) // 8 + 1, dtype=np.uint8), int(z["nbits"]))
        next_vid, snap = 0, 0
        snap_after = set(int(x) for x in z["snap_after"])
        for i in range(int(z["nops"])):
            op, a, b, ok = [int(x) for x in z["op_%d" % i]]
            pl = z["pl_%d" % i]
            if op == 0:
                keys = np.arange(next_vid, next_vid + b, dtype=np.int64)
                try:
                    g.add_keys(a, keys, pl)
                    got = 1
                except api.GammaHipError:
                    got = 0
                assert got == ok
                if ok:
                    next_vid += b
            elif op == 1:
                g.update(a, b, pl)
            elif op == 2:
                g.bitmap_set(pl.astype(np.int64), 1)
                g.delete(pl.astype(np.int64))
            elif op == 3:
                g.compact_if_need()
            if i in snap_after:
                for l in range(nlist):
                    ids, codes = g.get_list(l)
                    assert np.array_equal(ids, z["ids_%d_%d" % (snap, l)]), (snap, l)
                    assert np.array_equal(codes, z["codes_%d_%d" % (snap, l)]), (snap, l)
                caps = np.array([g.list_capacity(l) for l in range(nlist)])
                assert np.array_equal(caps, z["caps_%d" % snap]), snap
                snap += 1
    finally:
        g.close()


def test_moved_entries_are_skipped(case):
    """Update that moves a vector marks the old slot with bit 63 (kDelIdxMask); the scan must skip
    it (gamma_index_ivfpq.h:579-582) on both sides."""
    g = fixtures.load_hip(case)
    o = B.OracleIVFPQ(case["d"], case["nlist"], case["M"], 8, case["metric"])
    o.set_trained(case["cc"], case["pq"], None)
    for l in range(case["nlist"]):
        ids, codes = case["oracle"].get_list(l)
        if len(ids):
            o.add_keys(l, ids, codes)
    o.set_raw(case["base"])
    try:
        rng = np.random.default_rng(4)
        for vid in rng.choice(case["N"], size=200, replace=False):
            newl = int(rng.integers(0, case["nlist"]))
            code = rng.integers(0, 256, size=case["M"]).astype(np.uint8)
            B.lib().go_ivfpq_update_code(o.h, newl, int(vid), B._up(code))
            g.update(newl, int(vid), code)
        c2 = dict(case, oracle=o)
        for has_rank in (True, False):
            (D, I, _), (Dg, Ig) = run_both(c2, g, case["q"], 10, 16, 100, B.METRIC_L2, has_rank)
            compare_exact(D, I, Dg, Ig)
    finally:
        g.close()


def test_device_encode_matches_oracle(case, hip):
    """Add path on device (assign + residual + PQ argmin) == oracle == real faiss (golden)."""
    x = case["base"][:3000]
    for mode, n in ((0, 7), (1, 3000)):      # faiss rule: n < 20 exact, else GEMM form
        B.lib().go_set_assign_mode(mode)
        lo, co = case["oracle"].encode(x[:n])
        lg, cg = hip.encode(x[:n])
        assert np.array_equal(lo, lg) and np.array_equal(co, cg)
    B.lib().go_set_assign_mode(0)


# --------------------------------------------------------------------------- full size (C3)
@pytest.fixture(scope="module")
def c3():
    """BASELINE.json configs[2]: 1M x 128, nlist 4096, M 16; trained (faiss's IndexIVFPQ::train, gamma_hip_ivfpq_train) and
    encoded on the GPU; the oracle is loaded with the same lists for sampled parity."""
    N, d, nlist, M = 1000000, 128, 4096, 16
    base = synth.sift_like(N, d=d, seed=1234)
    g = api.GammaHip(0)
    # trained the way a Gamma table's Indexing() trains it: IndexIVFPQ::train on the first nlist * 64 vectors, on the device
    cc, pq = g.ivfpq_train(base[:nlist * 64], nlist, M)
    g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
    g.ivfpq_set_trained(cc, pq, None)
    for i0 in range(0, N, 250000):          # Add path in batches: exercises list growth
        g.add(base[i0:i0 + 250000], i0)
    g.raw_init(d)
    g.raw_append(base)
    yield dict(N=N, d=d, nlist=nlist, M=M, base=base, cc=cc, pq=pq, g=g)
    g.close()


def test_c3_properties(c3):
    g, base, N = c3["g"], c3["base"], c3["N"]
    assert sum(g.list_size(l) for l in range(c3["nlist"])) == N
    q = synth.sift_like(1024, d=128, seed=4321)
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True,
                          min_score=0.0, max_score=1e30, coarse_mode=1)
    D, I = g.ivfpq_search(q, 10, args)
    # sorted, unique, in range
    assert (np.diff(D, axis=1) >= 0).all()
    assert (I >= 0).all() and (I < N).all()
    assert all(len(set(r.tolist())) == 10 for r in I)
    # idempotent
    D2, I2 = g.ivfpq_search(q, 10, args)
    assert D.tobytes() == D2.tobytes() and np.array_equal(I, I2)
    # batch split invariance (per-query results do not depend on the batch composition)
    parts = [g.ivfpq_search(q[i:i + 256], 10, args) for i in range(0, 1024, 256)]
    assert np.concatenate([p[0] for p in parts]).tobytes() == D.tobytes()
    assert np.array_equal(np.concatenate([p[1] for p in parts]), I)
    # re-ranked distances are the exact ones
    ex = ((base[I[:50].ravel()] - np.repeat(q[:50], 10, axis=0)) ** 2).sum(1).reshape(50, 10)
    assert np.array_equal(ex.astype(np.float32), D[:50])     # integer-valued data: exact in fp32
    # self-queries come back first with distance 0
    Ds, Is = g.ivfpq_search(base[5000:5064], 10, args)
    assert (Is[:, 0] == np.arange(5000, 5064)).mean() > 0.95 and (Ds[Is[:, 0] == np.arange(5000, 5064), 0] == 0).all()
    # recall@10 >= 0.95 against exact flat search on the GPU
    Df, If = g.flat_search(q[:200], 10, api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30))
    rec = np.mean([len(set(I[i].tolist()) & set(If[i].tolist())) / 10.0 for i in range(200)])
    assert rec >= 0.95, rec


@pytest.fixture(scope="module")
def c3_oracle(c3):
    g = c3["g"]
    o = B.OracleIVFPQ(c3["d"], c3["nlist"], c3["M"], 8, B.METRIC_L2, bucket_init_size=4000)
    o.set_trained(c3["cc"], c3["pq"], g.ivfpq_table())
    for l in range(c3["nlist"]):
        ids, codes = g.get_list(l)
        if len(ids):
            o.add_keys(l, ids, codes)
    o.set_raw(c3["base"])
    return o


def test_c3_sampled_oracle_parity(c3, c3_oracle):
    g, o = c3["g"], c3_oracle
    q = synth.sift_like(96, d=128, seed=999)
    case = dict(oracle=o)
    for has_rank, cm in ((True, 1), (False, 1), (True, 0)):
        (D, I, st), (Dg, Ig) = run_both(case, g, q, 10, 32, 200, B.METRIC_L2, has_rank, coarse_mode=cm,
                                        ctx_kw=dict(min_score=0.0, max_score=1e30))
        sg = g.last_stages(len(q), 32, 200)
        assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
        compare_exact(D, I, Dg, Ig)


def test_c3_headline_call_matches_oracle(c3, c3_oracle):
    """The call bench.py times, pinned: ONE 16384-query Search on the C3 index (nprobe 32, recall_num 200, k 10), i.e.
    the matrix-free GEMM-form coarse quantizer (k_coarse_fused + the heap replay of its tied rows), the bounded scan
    with eight probes per workgroup, k_select_final, k_rerank_topk and the tie replay.  A sample of the batch --
    every 67th query plus queries whose result holds equal distances -- against the oracle: coarse assignment byte
    for byte, recall-stage tables, and the final labels at EVERY rank (exact ties are the default; the oracle's
    heaps are pinned against the compiled faiss, tests/test_oracle_golden.py)."""
    g, o = c3["g"], c3_oracle
    nq, P, R, k = 16384, 32, 200, 10
    q = synth.sift_like(nq, d=128, seed=4321)
    ctx = B.make_ctx(min_score=0.0, max_score=1e30)
    for has_rank in (True, False):
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=has_rank, min_score=0.0,
                              max_score=1e30, coarse_mode=-1)
        g.tie_stats(reset=True)
        Dg, Ig = g.ivfpq_search(q, k, args)
        ts = g.tie_stats()
        sg = g.last_stages(nq, P, R)
        tied = np.nonzero((np.diff(Dg, axis=1) == 0).any(axis=1))[0]
        rows = np.unique(np.concatenate([np.arange(0, nq, 67), tied[:80]]))
        D, I, st = o.search(q[rows], k, P, recall_num=R, has_rank=has_rank, metric=B.METRIC_L2, ctx=ctx, coarse_mode=1,
                            want_stages=True)
        assert sg["coarse_dis"][rows].tobytes() == st["coarse_dis"].tobytes()
        assert np.array_equal(sg["coarse_idx"][rows], st["coarse_idx"])
        sub = dict(recall_dis=sg["recall_dis"][rows], recall_ids=sg["recall_ids"][rows])
        assert compare_search_exact(D, I, st, Dg[rows], Ig[rows], sub) == 0
        compare_exact(D, I, Dg[rows], Ig[rows])
        assert ts["coarse_rows"] > 0 and ts["replayed"] > 0   # the sample really went through the replays
        if has_rank:
            assert len(tied) > 0


def test_c2_flat_full_size(c3):
    """BASELINE.json configs[1]: Flat L2 over 1M x 128, 1024 queries per call, k = 100."""
    import time
    g, base, N = c3["g"], c3["base"], c3["N"]
    q = synth.sift_like(1024, d=128, seed=77)
    args = api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30)
    g.flat_search(q[:64], 100, args)
    t0 = time.perf_counter()
    D, I = g.flat_search(q, 100, args)
    dt = time.perf_counter() - t0
    print("C2 flat: 1024 queries x 1M x 128, k=100: %.1f ms (%.0f queries/s, host buffers)" % (dt * 1e3, 1024 / dt))
    assert (np.diff(D, axis=1) >= 0).all()
    assert (I >= 0).all() and (I < N).all()
    assert all(len(set(r.tolist())) == 100 for r in I)
    # distances are the exact fvec_L2sqr values (integer-valued data: exact in fp32)
    ex = ((base[I[:40].ravel()] - np.repeat(q[:40], 100, axis=0)) ** 2).sum(1).reshape(40, 100)
    assert np.array_equal(ex.astype(np.float32), D[:40])
    # nothing closer was missed: the k-th distance bounds every unreturned vector (sampled)
    for i in range(0, 40, 8):
        dall = ((base - q[i]) ** 2).sum(1)
        assert np.partition(dall, 99)[99] == D[i, 99]
    # the oracle (reference flat loop restated) over the full base: every query of the batch with an exact-distance tie among
    # its first k + 1 results (the rows the tie replay rewrites: integer-valued data has many) up to 48 of them, plus queries
    # spread over the batch -- at least 64 in all (6 of 1024 was the whole oracle sample)
    D101, _ = g.flat_search(q, 101, args)
    tied = np.nonzero((np.diff(D101, axis=1) == 0).any(axis=1))[0]
    sel = np.unique(np.concatenate([tied[:48], np.arange(0, 1024, 16)]))
    assert len(sel) >= 64
    Do, Io = B.flat_search(base, np.ascontiguousarray(q[sel]), 100, B.METRIC_L2, B.make_ctx(min_score=0.0, max_score=1e30))
    compare_exact(Do, Io, D[sel], I[sel])
    print("C2 oracle sample: %d queries, %d of them with a tie among the first 101 distances (%d in the batch)" % (len(sel), min(len(tied), 48), len(tied)))
    # inner product over the same store
    argi = api.SearchArgs(metric=api.METRIC_IP, **WIDE)
    Di, Ii = g.flat_search(q[:6], 100, argi)
    Do, Io = B.flat_search(base, q[:6], 100, B.METRIC_IP, B.make_ctx(**WIDE))
    compare_exact(Do, Io, Di, Ii)


@pytest.mark.parametrize("shape", ["m8", "m64"])
def test_scan_bound_parity_at_batch_size(case, shape):
    """512+ queries x 32 probes switch the threshold pre-filter on (4 probes per workgroup): full
    parity against the oracle, with and without filters, both metrics, re-rank on and off.
    m64: the 64-byte-code variant of the scan (two codes per thread), lists of ~600."""
    if shape == "m64":
        case = fixtures.trained_case(d=128, nlist=32, M=64, N=20000, nq=24, metric=B.METRIC_L2)
    g = fixtures.load_hip(case)
    try:
        q = synth.sift_like(530, d=case["d"], seed=4242)
        N = case["N"]
        rng = np.random.default_rng(8)
        dead = rng.choice(N, size=N // 4, replace=False)
        bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        keep = rng.choice(N, size=N // 2, replace=False)
        for del_bm, rdocs in ((None, None), (bm, [keep])):
            if del_bm is not None:
                g.bitmap_upload(del_bm, N)
                case["oracle"].set_docids_bitmap(del_bm)
            for metric in (B.METRIC_L2, B.METRIC_IP):
                for has_rank in (True, False):
                    (D, I, st), (Dg, Ig) = run_both(case, g, q, 10, 32, 100, metric, has_rank, coarse_mode=1,
                                                    del_bitmap=del_bm, range_docs=rdocs)
                    sg = g.last_stages(len(q), 32, 100)
                    assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
                    compare_search_exact(D, I, st, Dg, Ig, sg)
    finally:
        case["oracle"].set_docids_bitmap(np.zeros((case["N"] >> 3) + 1, dtype=np.uint8))
        g.close()


@pytest.mark.parametrize("P", [80, 128])
def test_scan_bound_with_more_than_64_probes(P):
    """nprobe 80 is the reference's default (gamma_index_ivfpq.h:629-673): the threshold pre-filter and its
    selection kernel cover up to 128 probes per query."""
    case = fixtures.trained_case(d=32, nlist=160, M=8, N=30000, nq=24, metric=B.METRIC_L2)
    g = fixtures.load_hip(case)
    try:
        q = synth.sift_like(300, d=case["d"], seed=99)
        for metric in (B.METRIC_L2, B.METRIC_IP):
            for has_rank in (True, False):
                (D, I, st), (Dg, Ig) = run_both(case, g, q, 10, P, 100, metric, has_rank, coarse_mode=1)
                sg = g.last_stages(len(q), P, 100)
                assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
                compare_search_exact(D, I, st, Dg, Ig, sg)
    finally:
        g.close()


def test_coarse_ties_at_the_nprobe_boundary_follow_the_reference_heap():
    """Centroids duplicated in pairs: with an odd nprobe the nprobe-th and the next coarse distance are always
    equal, and WHICH of the two lists is probed is decided by faiss's heap (HeapResultHandler).  The device
    redoes such rows the way the heap does (k_coarse_heap_fix): the probed sets must be identical, for the
    exact (nq < 20) and the GEMM-form coarse path, and the searches must agree downstream."""
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    cc = case["cc"].copy()
    cc[1::2] = cc[0::2]
    base = case["base"][:8000]
    o = B.OracleIVFPQ(case["d"], case["nlist"], case["M"], 8, B.METRIC_L2)
    o.set_trained(cc, case["pq"], None)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, api.METRIC_L2, 1000)
        g.ivfpq_set_trained(cc, case["pq"], None)
        g.set_exact_ties(True)          # opt-in: a flagged row is a sequential walk
        g.raw_init(case["d"])
        g.raw_append(base)
        B.lib().go_set_assign_mode(1)
        g.add(base, 0)
        assert o.add(base)
        B.lib().go_set_assign_mode(0)
        o.set_raw(base)
        q = synth.sift_like(300, d=case["d"], seed=4)
        n_split = 0
        for nq, P in ((5, 7), (24, 7), (300, 7), (300, 31), (24, 32), (300, 64)):
            ctx = B.make_ctx(**WIDE)
            D, I, st = o.search(q[:nq], 10, P, recall_num=60, has_rank=True, metric=B.METRIC_L2, ctx=ctx,
                                coarse_mode=-1, want_stages=True)
            Dg, Ig = g.ivfpq_search(q[:nq], 10, api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=60,
                                                              has_rank=True, coarse_mode=-1, **WIDE))
            sg = g.last_stages(nq, P, 60)
            assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
            for a, b in zip(st["coarse_idx"], sg["coarse_idx"]):
                assert set(a.tolist()) == set(b.tolist())
                n_split += (a[-1] ^ 1) not in a.tolist() if P < 64 else 0
            compare_search_exact(D, I, st, Dg, Ig, sg)
        assert n_split > 100          # the boundary really did split pairs
    finally:
        B.lib().go_set_assign_mode(0)
        g.close()


def test_scan_bound_fallback_paths():
    """The threshold pre-filter of the scan must hand a query over to the unfiltered selection when
    it has no usable bound: (a) mass ties -- every candidate within the bound, survivor slices
    overflow; (b) a first probe group with fewer than recall_num valid candidates (90% deleted)."""
    d, nlist, M, N = 32, 64, 8, 24000
    rng = np.random.default_rng(3)
    distinct = synth.sift_like(60, d=d, seed=77)
    base = distinct[rng.integers(0, 60, size=N)].copy()          # 60 distinct vectors, 400 copies each
    base[:2000] = synth.sift_like(2000, d=d, seed=78)            # plus some ordinary ones
    # 520 queries x 32 probes: enough workgroups for 4 probes per group, i.e. the pre-filter is on
    q = np.concatenate([distinct[:60], synth.sift_like(460, d=d, seed=79)])
    from tests import lloyd as train
    cc, pq = train.train_ivfpq(base[:6000], nlist, M, niter=6, pq_niter=8, seed=9, device="cpu")
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2, bucket_init_size=4000)
    o.set_trained(cc, pq, None)
    B.lib().go_set_assign_mode(1)
    assert o.add(base)
    B.lib().go_set_assign_mode(0)
    o.set_raw(base)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 4000)
        g.ivfpq_set_trained(cc, pq, None)
        g.add(base, 0)
        g.raw_init(d)
        g.raw_append(base)
        case = dict(oracle=o)
        dead = rng.choice(N, size=int(N * 0.9), replace=False)
        bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        for del_bm in (None, bm):
            if del_bm is not None:
                g.bitmap_upload(del_bm, N)
                o.set_docids_bitmap(del_bm)
            for has_rank in (True, False):
                for metric in (B.METRIC_L2, B.METRIC_IP):
                    (D, I, st), (Dg, Ig) = run_both(case, g, q, 10, 32, 120, metric, has_rank, coarse_mode=1,
                                                    del_bitmap=del_bm)
                    sg = g.last_stages(len(q), 32, 120)
                    compare_search_exact(D, I, st, Dg, Ig, sg)
    finally:
        g.close()


def test_concurrent_search_and_add_on_one_handle(case):
    """Search is re-entrant for the engine's client threads while the indexing thread appends
    (SURVEY §8b threading): calls serialise on the handle, results stay exact."""
    import threading
    g = api.GammaHip(0)
    o = case["oracle"]
    try:
        g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
        g.ivfpq_set_trained(case["cc"], case["pq"], None)
        g.raw_init(case["d"])
        base, q = case["base"], case["q"]
        half = len(base) // 2
        g.raw_append(base[:half])
        g.add(base[:half], 0)
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True, coarse_mode=0, **WIDE)
        errors, results = [], {}

        def searcher(t):
            try:
                for it in range(6):
                    D, I = g.ivfpq_search(q, 10, args)
                    assert (np.diff(D, axis=1) >= 0).all() and (I < len(base)).all()
                    results[t] = (D, I)
            except Exception as e:   # noqa: BLE001
                errors.append(e)

        def adder():
            try:
                for i0 in range(half, len(base), 1000):
                    g.raw_append(base[i0:i0 + 1000])
                    g.add(base[i0:i0 + 1000], i0)
            except Exception as e:   # noqa: BLE001
                errors.append(e)

        th = [threading.Thread(target=searcher, args=(t,)) for t in range(4)] + [threading.Thread(target=adder)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        # everything is in: the final state answers exactly like the oracle that holds the same lists
        D, I = o.search(q, 10, 8, recall_num=100, has_rank=True, metric=B.METRIC_L2, ctx=B.make_ctx(**WIDE),
                        coarse_mode=0)
        Dg, Ig = g.ivfpq_search(q, 10, args)
        compare_exact(D, I, Dg, Ig)
    finally:
        g.close()


def test_concurrent_small_searches_are_combined_and_exact(case, hip):
    """Many client threads with one or a few queries each (the engine's serving pattern): requests that
    arrive while the GPU is busy share a batch.  Every result must be bit-identical to the same call made
    alone -- including the coarse path, which each request's OWN size selects (exact below 20 queries)."""
    import threading
    q = synth.sift_like(480, d=case["d"], seed=515)
    variants = [
        (1, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True, **WIDE), 10),
        (3, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True, **WIDE), 10),
        (1, api.SearchArgs(metric=api.METRIC_IP, nprobe=12, recall_num=60, has_rank=False, **WIDE), 5),
        (24, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True, **WIDE), 10),   # GEMM-form coarse
    ]
    # requests with their OWN range filters share batches too (one filter-table entry per request)
    frng = np.random.default_rng(77)
    N = case["N"]
    for sel, not_in in ((0.5, False), (0.1, False), (0.3, True)):
        docs = np.sort(frng.choice(N, size=int(N * sel), replace=False))
        variants.append((1, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True,
                                           range_filters=[api.make_range_filter(docs, b_not_in=not_in)], **WIDE), 10))
    variants.append((2, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True,
                                       range_filters=[api.make_range_filter(np.arange(100, 9000)),
                                                      api.make_range_filter(np.arange(5000, 19000))], **WIDE), 10))
    # flat searches (the brute-force fallback of the plugins) from the same clients
    flat_variants = {len(variants): True, len(variants) + 1: True}
    variants.append((1, api.SearchArgs(metric=api.METRIC_L2, **WIDE), 10))
    variants.append((2, api.SearchArgs(metric=api.METRIC_IP, **WIDE), 7))

    def call(v, xb, k, args):
        return hip.flat_search(xb, k, args) if v in flat_variants else hip.ivfpq_search(xb, k, args)

    # the answers of the calls made one at a time
    want = {}
    for v, (n, args, k) in enumerate(variants):
        for i0 in range(0, 480 - n + 1, n):
            want[(v, i0)] = call(v, q[i0:i0 + n], k, args)
    errors = []

    def client(t):
        try:
            rng = np.random.default_rng(t)
            for it in range(60):
                v = int(rng.integers(0, len(variants)))
                n, args, k = variants[v]
                i0 = int(rng.integers(0, (480 - n) // n + 1)) * n
                D, I = call(v, q[i0:i0 + n], k, args)
                Dw, Iw = want[(v, i0)]
                assert D.tobytes() == Dw.tobytes() and np.array_equal(I, Iw), (t, it, v, i0)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=client, args=(t,)) for t in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


_FILTER_BITMAP_MODES = ("0", "1", None)


class _filter_bitmap:
    """GAMMA_HIP_FILTER_BITMAP for the calls inside: "0" clauses evaluated per scored code, "1" once per document into a
    bitmap (k_filter_bitmap), None: the library's own estimate decides."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        import os
        self.old = os.environ.get("GAMMA_HIP_FILTER_BITMAP")
        if self.mode is None:
            os.environ.pop("GAMMA_HIP_FILTER_BITMAP", None)
        else:
            os.environ["GAMMA_HIP_FILTER_BITMAP"] = self.mode

    def __exit__(self, *a):
        import os
        if self.old is None:
            os.environ.pop("GAMMA_HIP_FILTER_BITMAP", None)
        else:
            os.environ["GAMMA_HIP_FILTER_BITMAP"] = self.old


def test_field_filters_on_device(case):
    """Scalar range filters evaluated on device columns == the same selection handed over as a
    host-built RangeQueryResult bitmap (reference semantics: IsInRange<T>, AND of the clauses,
    include_lower / include_upper, docs beyond the column do not match)."""
    N = case["N"]
    rng = np.random.default_rng(17)
    price = rng.integers(0, 1000, size=N).astype(np.int64)
    score = rng.random(N - 500)                      # float64, 500 docs short
    stock = rng.integers(-5, 5, size=N).astype(np.int32)
    weight = rng.random(N).astype(np.float32)
    g = fixtures.load_hip(case)       # own handle: no delete bitmap left over from other tests
    g.field_append(1, price[:N // 2])
    g.field_append(1, price[N // 2:])                # grows
    g.field_append(2, score)
    g.field_append(3, stock)
    g.field_append(4, weight)
    assert g.field_count(1) == N and g.field_count(2) == N - 500
    g.field_update(1, 7, np.array([price[7] + 1000], dtype=np.int64))
    price[7] += 1000
    sc = np.concatenate([score, np.full(500, np.nan)])
    w_lo, w_hi = np.float32(0.25), np.float32(0.75)
    cases = [
        ([(1, 100, 300, True, True)], (price >= 100) & (price <= 300)),
        ([(1, 100, 300, False, False)], (price > 100) & (price < 300)),
        ([(1, 100, 300, True, False), (2, 0.2, 0.9, False, True)],
         (price >= 100) & (price < 300) & (sc > 0.2) & (sc <= 0.9)),
        ([(3, -2, 2, True, True), (4, float(w_lo), float(w_hi), True, False)],
         (stock >= -2) & (stock <= 2) & (weight >= w_lo) & (weight < w_hi)),
        ([(1, 5000, 6000, True, True)], np.zeros(N, bool)),
    ]
    q = case["q"]
    for filters, mask in cases:
        docs = np.nonzero(mask)[0]
        for has_rank in (True, False):
            (D, I, st), _ = run_both(case, g, q, 10, 8, 100, B.METRIC_L2, has_rank, range_docs=[docs])
            args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=has_rank,
                                  coarse_mode=0, field_filters=filters, **WIDE)
            for fb in _FILTER_BITMAP_MODES:   # clauses per scored code / once per document into a bitmap / by estimate
                with _filter_bitmap(fb):
                    Dg, Ig = g.ivfpq_search(q, 10, args)
                compare_exact(D, I, Dg, Ig)
        Df, If = B.flat_search(case["base"], q[:8], 10, B.METRIC_L2,
                               B.make_ctx(range_filters=[B.make_range_filter(docs)], **WIDE))
        for fb in _FILTER_BITMAP_MODES:
            with _filter_bitmap(fb):
                Dg, Ig = g.flat_search(q[:8], 10, api.SearchArgs(metric=api.METRIC_L2, field_filters=filters, **WIDE))
            compare_exact(Df, If, Dg, Ig)
    with pytest.raises(api.GammaHipError):      # unknown column
        g.ivfpq_search(q, 10, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, field_filters=[(99, 0, 1, True, True)],
                                             **WIDE))
    g.close()


def test_internal_chunking_keeps_the_whole_call_semantics(case):
    """Large calls are processed in chunks (ADC buffer budget).  faiss chooses the coarse path from the
    size of the whole call, so a trailing chunk of < 20 queries must still take the GEMM form."""
    g = fixtures.load_hip(case)
    try:
        q = synth.sift_like(45, d=case["d"], seed=31337)
        g.set_dist_budget(case["nlist"] * 4 * 20)        # room for 20 rows of the coarse matrix: chunks of 20, 20, 5
        (D, I, st), (Dg, Ig) = run_both(case, g, q, 10, 8, 100, B.METRIC_L2, True, coarse_mode=-1)
        compare_exact(D, I, Dg, Ig)
        # same for the Add path's quantizer->assign: 45 vectors in chunks of 20, 20, 5
        xs = case["base"][100:145]
        B.lib().go_set_assign_mode(1)
        lo, co = case["oracle"].encode(xs)
        B.lib().go_set_assign_mode(0)
        lg, cg = g.encode(xs)
        assert np.array_equal(lo, lg) and np.array_equal(co, cg)
        g.set_dist_budget(8 << 30)
        D2, I2 = g.ivfpq_search(q, 10, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, **WIDE))
        assert D2.tobytes() == Dg.tobytes() and np.array_equal(I2, Ig)
    finally:
        g.close()


def test_list_full_at_bucket_max_size(case):
    """A list cannot grow past bucket_max_size (RealTimeMemData::ExtendBucketIfNeed,
    realtime/realtime_mem_data.cc:383-474): AddKeys fails at the same batch on both sides and
    leaves the same list behind."""
    d, nlist, M = case["d"], case["nlist"], case["M"]
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2, bucket_init_size=8, bucket_max_size=40)
    o.set_trained(case["cc"], case["pq"], None)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 8, 40)
        g.ivfpq_set_trained(case["cc"], case["pq"], None)
        rng = np.random.default_rng(12)
        failed_o = failed_g = None
        for b in range(12):
            keys = np.arange(b * 7, b * 7 + 7, dtype=np.int64)
            codes = rng.integers(0, 256, size=(7, M)).astype(np.uint8)
            ok_o = o.add_keys(3, keys, codes)
            try:
                g.add_keys(3, keys, codes)
                ok_g = True
            except api.GammaHipError:
                ok_g = False
            assert ok_o == ok_g, b
            if not ok_o and failed_o is None:
                failed_o = failed_g = b
        assert failed_o is not None and failed_o == failed_g
        io, co = o.get_list(3)
        ig, cg = g.get_list(3)
        assert np.array_equal(io, ig) and np.array_equal(co, cg)
        assert o.list_capacity(3) == g.list_capacity(3) and len(io) <= 40
    finally:
        g.close()


def test_raw_write_is_idempotent():
    """gamma_hip_raw_write: rows at explicit positions; repeating or overlapping a write changes nothing, the
    count only grows, and a write that would leave a gap is refused."""
    from gamma_amd import api
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3000, 32)).astype(np.float32)
    g = api.GammaHip(0)
    g.raw_init(32)
    g.raw_write(0, x[:1000])
    g.raw_write(500, x[500:2000])      # overlaps and extends
    g.raw_write(0, x[:1000])           # repeated
    assert g.raw_count() == 2000
    with pytest.raises(api.GammaHipError):
        g.raw_write(2500, x[2500:])    # gap
    g.raw_write(2000, x[2000:])
    assert g.raw_count() == 3000
    q = x[::97].copy()
    args = api.SearchArgs(metric=api.METRIC_L2, min_score=-1e30, max_score=1e30)
    D, I = g.flat_search(q, 1, args)
    assert np.array_equal(I[:, 0], np.arange(0, 3000, 97)) and (D == 0).all()
    g.close()


def test_arena_repack_reclaims_abandoned_extents(case):
    """list growth and compaction abandon their old extents inside the arena; once the
    waste passes the threshold every list moves into a tight arena.  Contents, capacities and search results
    are unchanged by the move."""
    from gamma_amd import api
    o = case["oracle"]
    g = api.GammaHip(0)
    nlist, M = case["nlist"], case["M"]
    g.ivfpq_init(case["d"], nlist, M, 8, case["metric"], 16)   # tiny buckets: every list grows many times
    g.ivfpq_set_trained(case["cc"], case["pq"], None)
    g.set_repack_threshold(1 << 40)                             # off for now
    lists = [o.get_list(l) for l in range(nlist)]
    for rnd in range(4):                                        # interleaved appends: repeated growth
        for l in range(nlist):
            ids, cds = lists[l]
            lo, hi = len(ids) * rnd // 4, len(ids) * (rnd + 1) // 4
            if hi > lo:
                g.add_keys(l, ids[lo:hi], cds[lo:hi])
    st = g.arena_stats()
    assert st["waste"] > 0 and st["repacks"] == 0
    caps = [g.list_capacity(l) for l in range(nlist)]
    g.raw_init(case["d"])
    g.raw_append(case["base"])
    args = api.SearchArgs(metric=case["metric"], nprobe=8, recall_num=100, has_rank=True, min_score=-1e30,
                          max_score=1e30, coarse_mode=0)
    D0, I0 = g.ivfpq_search(case["q"], 10, args)
    g.set_repack_threshold(1)                                   # re-checks at once
    st2 = g.arena_stats()
    assert st2["repacks"] == 1 and st2["waste"] == 0 and st2["used"] == sum(caps)
    assert st2["cap"] < st["cap"] or st2["used"] < st["used"]
    for l in range(nlist):
        ids, cds = g.get_list(l)
        assert np.array_equal(ids, lists[l][0]) and np.array_equal(cds, lists[l][1])
        assert g.list_capacity(l) == caps[l]
    D1, I1 = g.ivfpq_search(case["q"], 10, args)
    assert D0.tobytes() == D1.tobytes() and np.array_equal(I0, I1)
    # appends after the move land in the new arena
    g.add_keys(0, np.array([10 ** 6], np.int64), lists[1][1][:1])
    ids, _ = g.get_list(0)
    assert ids[-1] == 10 ** 6 and len(ids) == len(lists[0][0]) + 1
    g.close()


def test_blas_corner_shapes_are_counted(case):
    """gamma_hip_blas_form_not_restated: a search / Add whose GEMM-form coarse call has a shape the restatement of MKL's
    sgemm_ does not cover is counted (never silent); the BASELINE shapes are not."""
    from gamma_amd import api
    g = fixtures.load_hip(case)
    try:
        args = api.SearchArgs(metric=case["metric"], nprobe=4, recall_num=20, has_rank=True, min_score=-1e30, max_score=1e30)
        q = synth.sift_like(4099, d=case["d"], seed=9)
        assert g.blas_form_not_restated(reset=True) == 0
        g.ivfpq_search(q[:4096], 5, args)
        g.ivfpq_search(q[:64], 5, args)
        g.ivfpq_search(q[:10], 5, args)          # below 20 queries: the exact form, no BLAS at all
        assert g.blas_form_not_restated() == 0
        g.ivfpq_search(q, 5, args)               # 4099 = 4096 + a 3-row remainder block
        assert g.blas_form_not_restated() == 1
        g.encode(q[:4097])
        assert g.blas_form_not_restated(reset=True) == 2
        assert g.blas_form_not_restated() == 0
    finally:
        g.close()


def _repack_script():
    """forced repacks between searches on a small index; prints the verify statistics and whether every state equalled the
    oracle's (run in-process and, with the fault injection, in a child process: the variable is read once per process)"""
    import json
    from gamma_amd import api
    from tests import fixtures
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    o = case["oracle"]
    nlist = case["nlist"]
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(case["d"], nlist, case["M"], 8, case["metric"], 16)
        g.ivfpq_set_trained(case["cc"], case["pq"], None)
        g.set_repack_threshold(1 << 40)
        g.raw_init(case["d"])
        g.raw_append(case["base"])
        lists = [o.get_list(l) for l in range(nlist)]
        args = api.SearchArgs(metric=case["metric"], nprobe=8, recall_num=100, has_rank=True, min_score=-1e30, max_score=1e30,
                              coarse_mode=0)
        Do, Io = o.search(case["q"], 10, 8, recall_num=100, has_rank=True, metric=case["metric"],
                          ctx=B.make_ctx(min_score=-1e30, max_score=1e30), coarse_mode=0)
        ok, modes = True, []
        for rnd in range(4):
            for l in range(nlist):
                ids, cds = lists[l]
                lo, hi = len(ids) * rnd // 4, len(ids) * (rnd + 1) // 4
                if hi > lo:
                    g.add_keys(l, ids[lo:hi], cds[lo:hi])
            g.set_repack_threshold(1)          # repack now
            g.set_repack_threshold(1 << 40)
            modes.append(bool(g.arena_growth()["mapped"]))
            for l in range(0, nlist, 7):
                gi, gc = g.get_list(l)
                hi = len(lists[l][0]) * (rnd + 1) // 4
                ok = ok and np.array_equal(gi, lists[l][0][:hi]) and np.array_equal(gc, lists[l][1][:hi])
        D, I = g.ivfpq_search(case["q"], 10, args)
        ok = ok and D.tobytes() == Do.tobytes() and np.array_equal(I, Io)
        return dict(stats=g.repack_verify_stats(), repacks=g.arena_stats()["repacks"], ok=bool(ok), in_place=modes)
    finally:
        g.close()


def test_arena_repack_publishes_only_what_reads_back():
    """VERDICT r4 #7 / ADVICE r4 (medium): a repack's new version of the list tables is published only after the target,
    read back through its new mapping in a launch of its own, equals the source (per-list checksums of ids + codes).
    Every forced repack is verified; none differs.  With the fault injection (a zeroed entry in the first read-back, child
    process) the difference is COUNTED, the old version stays, the move is repeated into ordinary allocations -- the arena
    leaves virtual memory management -- and every state is still the oracle's (realtime_mem_data.cc:426-474)."""
    import json
    import subprocess
    import sys
    r = _repack_script()
    assert r["ok"] and r["repacks"] >= 1 and r["stats"]["verified"] >= r["repacks"] and r["stats"]["failures"] == 0, r
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAMMA_HIP_FAULT_REPACK="1")
    c = subprocess.run([sys.executable, "-c", "import json; from tests.test_gpu_more import _repack_script; print('RES', json.dumps(_repack_script()))"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stdout[-2000:] + c.stderr[-3000:]
    rf = json.loads([l for l in c.stdout.splitlines() if l.startswith("RES ")][-1][4:])
    assert rf["ok"] and rf["stats"]["failures"] == 1 and rf["stats"]["verified"] == rf["repacks"] + 1, rf
    if r["in_place"][-1]:                      # (a runtime without virtual memory management starts outside it)
        assert not rf["in_place"][-1], rf      # the arena left virtual memory management at the failed read-back


def test_add_keys_batch_counts_superseded_slots(case):
    """ids with bit 63 (slots superseded by an Update, as a dump holds them) loaded through
    the batch entry point must make the scan read the ids -- such a slot is never returned."""
    from gamma_amd import api
    o = case["oracle"]
    g = api.GammaHip(0)
    g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
    g.ivfpq_set_trained(case["cc"], case["pq"], None)
    lists, counts, vids, codes = [], [], [], []
    moved = []
    for l in range(case["nlist"]):
        ids, cds = o.get_list(l)
        if len(ids) == 0:
            continue
        ids = ids.copy()
        if len(ids) > 3:
            moved.append(int(ids[1]))
            ids[1] |= np.int64(-2 ** 63)
        lists.append(l); counts.append(len(ids)); vids.append(ids); codes.append(cds)
    g.add_keys_batch(lists, counts, np.concatenate(vids), np.concatenate(codes))
    g.raw_init(case["d"])
    g.raw_append(case["base"])
    args = api.SearchArgs(metric=case["metric"], nprobe=16, recall_num=100, has_rank=False, min_score=-1e30,
                          max_score=1e30, coarse_mode=0)
    D, I = g.ivfpq_search(case["q"], 20, args)
    assert (I >= -1).all() and not np.isin(I, moved).any()
    # a list named twice in one batch reserves for the sum of its parts
    g2 = api.GammaHip(0)
    g2.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 8)
    g2.ivfpq_set_trained(case["cc"], case["pq"], None)
    ids, cds = o.get_list(lists[0])
    n = len(ids)
    g2.add_keys_batch([lists[0], lists[0]], [n // 2, n - n // 2], ids, cds)
    gi, gc = g2.get_list(lists[0])
    assert np.array_equal(gi, ids) and np.array_equal(gc, cds)
    g.close()
    g2.close()


def _term_mask(doc_items, items, op):
    want = set(items)
    if op == 1:
        return np.array([bool(want & set(d)) for d in doc_items])
    if op == 2:
        return np.array([not (want & set(d)) for d in doc_items])
    return np.array([want <= set(d) for d in doc_items])


def test_term_filters_on_device(case):
    """Term (tag) filters over dictionary-encoded item lists in HBM == the same selection handed over as a
    docid bitmap.  And = every item present (index/impl/gpu/gamma_index_ivfpq_gpu.cc:728-760), Or = any,
    Not = none of them (table/field_range_index.cc:1052-1056, SetNotIn); an item no doc carries is -1;
    docs beyond the column match no clause (as for numeric columns); clauses AND with each other and with numeric clauses."""
    N = case["N"]
    rng = np.random.default_rng(23)
    n_tags = 12
    counts = rng.integers(0, 4, size=N - 300).astype(np.int32)      # 300 docs short
    items = rng.integers(0, n_tags, size=int(counts.sum())).astype(np.int32)
    price = rng.integers(0, 1000, size=N).astype(np.int64)
    g = fixtures.load_hip(case)
    off = np.concatenate([[0], np.cumsum(counts)])
    doc_items = [items[off[i]:off[i + 1]].tolist() for i in range(N - 300)]
    half = (N - 300) // 2
    g.term_append(5, doc_items[:half])
    g.term_append(5, doc_items[half:])                               # grows
    g.field_append(1, price)
    assert g.term_count(5) == N - 300
    doc_items += [[] for _ in range(300)]
    cases = [
        ([(5, 1, [3])], []),
        ([(5, 0, [3, 7])], []),
        ([(5, 1, [1, 2, 9])], []),
        ([(5, 2, [0, 4])], []),
        ([(5, 1, [2, -1])], []),                                   # -1: an item no doc carries
        ([(5, 0, [2, -1])], []),                                   # And with an unknown item: nothing
        ([(5, 1, [6]), (5, 2, [8])], []),
        ([(5, 1, [1, 5])], [(1, 200, 700, True, False)]),
    ]
    q = case["q"]
    for terms, fields in cases:
        mask = np.ones(N, bool)
        for fid, op, its in terms:
            mask &= _term_mask(doc_items, its, op)
            mask[N - 300:] = False          # beyond the column: no clause matches, Not included
        for fid, lo, hi, il, iu in fields:
            mask &= (price >= lo) & (price < hi)
        docs = np.nonzero(mask)[0]
        for has_rank in (True, False):
            (D, I, st), _ = run_both(case, g, q, 10, 8, 100, B.METRIC_L2, has_rank, range_docs=[docs])
            args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=has_rank, coarse_mode=0,
                                  term_filters=terms, field_filters=fields or None, **WIDE)
            for fb in _FILTER_BITMAP_MODES:
                with _filter_bitmap(fb):
                    Dg, Ig = g.ivfpq_search(q, 10, args)
                compare_exact(D, I, Dg, Ig)
        Df, If = B.flat_search(case["base"], q[:8], 10, B.METRIC_L2,
                               B.make_ctx(range_filters=[B.make_range_filter(docs)], **WIDE))
        for fb in _FILTER_BITMAP_MODES:
            with _filter_bitmap(fb):
                Dg, Ig = g.flat_search(q[:8], 10, api.SearchArgs(metric=api.METRIC_L2, term_filters=terms,
                                                                 field_filters=fields or None, **WIDE))
            compare_exact(Df, If, Dg, Ig)
    with pytest.raises(api.GammaHipError):      # unknown column
        g.ivfpq_search(q, 10, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, term_filters=[(77, 1, [1])], **WIDE))
    g.close()


@pytest.mark.parametrize("d,nlist,P,nq", [(128, 4096, 32, 4500), (32, 2048, 8, 4100), (64, 4100, 64, 4200),
                                          (96, 2304, 20, 8200), (128, 4096, 1, 4096), (32, 8192, 32, 4100),
                                          (32, 16384, 32, 4100), (32, 32768, 24, 4100), (32, 16384, 64, 4100), (64, 8192, 40, 4100)])
def test_fused_coarse_matches_matrix_path_and_oracle(d, nlist, P, nq):
    """csrc/coarse.hip (sample -> bound -> filtered GEMM epilogue -> merge) == the distance-matrix path, bit for
    bit, also when survivor lists overflow and queries go through the repair kernel (list_cap 1: every query;
    40: some), and == the oracle's GEMM-form knn_L2sqr (faiss:utils/distances.cpp:215-296) up to ties.  The larger
    nlist take 16 strips and a 1024 / 2048-column sample (coarse_fused_shape); nprobe > 32 bounds with two minima per lane."""
    import torch
    rng = np.random.default_rng(d + nlist + P)
    cc = (rng.standard_normal((nlist, d)) * 20).astype(np.float32)
    cc[5] = cc[600]                                   # equal distances: (distance, list number) order in both paths
    cc[nlist - 1] = cc[700]
    x = cc[rng.integers(0, nlist, nq)] + (rng.standard_normal((nq, d)) * 12).astype(np.float32)
    x[3] = cc[5]                                      # distance 0 twice
    M = d // 8
    pq = rng.standard_normal((M, 256, d // M)).astype(np.float32)
    g = api.GammaHip(0)
    g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 100)
    g.ivfpq_set_trained(cc, pq, None)
    dev = torch.device("cuda", 0)
    dx = torch.from_numpy(x).to(dev)
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, coarse_mode=1, **WIDE)

    def run():
        cd = torch.empty((nq, P), dtype=torch.float32, device=dev)
        ci = torch.empty((nq, P), dtype=torch.int32, device=dev)
        g.ivfpq_coarse_device(dx.data_ptr(), nq, args, cd.data_ptr(), ci.data_ptr())
        g.synchronize()
        return cd.cpu().numpy(), ci.cpu().numpy()

    try:
        g.set_coarse_fused(False)
        D0, I0 = run()
        for cap in (128, 40, 1):
            g.set_coarse_fused(True, cap)
            D1, I1 = run()
            assert D0.tobytes() == D1.tobytes(), cap
            assert np.array_equal(I0, I1), cap
        Do, Io = B.knn_L2sqr(x[:600], cc, P, mode=1)
        compare_exact(Do, Io, D0[:600], I0[:600].astype(np.int64))
    finally:
        g.close()


@pytest.mark.parametrize("d,M,nlist,metric", [(32, 8, 64, B.METRIC_L2), (128, 16, 256, B.METRIC_L2),
                                              (64, 32, 64, B.METRIC_L2), (32, 8, 64, B.METRIC_IP),
                                              (100, 20, 64, B.METRIC_L2), (100, 25, 96, B.METRIC_IP),
                                              (768, 64, 64, B.METRIC_IP)])
def test_small_batch_path_is_the_regular_chain(d, M, nlist, metric):
    """Calls of up to 512 queries run as four or five fused kernels (gamma_hip_search.cpp ivfpq_small); every output and every stage
    table must equal the regular chain's, byte for byte: with / without re-rank, recall_num above and below the
    candidate count, k > candidates, deleted docs, a score window, and the oracle for good measure."""
    case = fixtures.trained_case(d=d, nlist=nlist, M=M, N=20000 if d < 512 else 6000, nq=64, metric=metric)
    g = fixtures.load_hip(case)
    rng = np.random.default_rng(d + M)
    hip_metric = api.METRIC_L2 if metric == B.METRIC_L2 else api.METRIC_IP
    N = case["N"]
    dead = rng.choice(N, N // 10, replace=False)
    try:
        for step in range(2):
            if step == 1:
                bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
                np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
                g.bitmap_upload(bm, N)
            big = synth.sift_like(300, d=d, seed=4242)
            for nq in (1, 2, 5, 8, 13, 16, 20, 37, 64, 300):
                q = case["q"][:nq] if nq <= 64 else big
                for has_rank in (True, False):
                    for P, R, k, win in ((8, 100, 10, None), (1, 50, 10, None), (32, 1000, 100, None),
                                         (4, 20, 30, None), (16, 200, 10, True), (64, 300, 10, None),
                                         (min(100, nlist), 250, 10, None)):   # the reference's default nprobe is 80
                        kw = dict(WIDE)
                        if win:   # a window that cuts on both sides, whatever the metric and the scale of the data
                            Dw, _ = g.ivfpq_search(q, k, api.SearchArgs(metric=hip_metric, nprobe=P, recall_num=R,
                                                                        has_rank=has_rank, **WIDE))
                            fin = Dw[np.isfinite(Dw) & (np.abs(Dw) < 1e37)]
                            kw = dict(min_score=float(np.quantile(fin, 0.2)), max_score=float(np.quantile(fin, 0.8)))
                        # coarse_mode 0 = exact coarse distances whatever the batch (what a combined batch of
                        # single-query requests asks for): beyond 16 queries the regular chain's kernel feeds the chain
                        for cm in ((-1, 0) if (has_rank and nq in (20, 37, 300)) else (-1,)):
                            args = api.SearchArgs(metric=hip_metric, nprobe=P, recall_num=R, has_rank=has_rank,
                                                  coarse_mode=cm, **kw)
                            g.set_small_path(False)
                            D0, I0 = g.ivfpq_search(q, k, args)
                            s0 = g.last_stages(nq, P, max(R, k))
                            # 1: automatic; 3: long-row selection in two levels forced (three slices at most)
                            for mode in ((1, 3) if nq in (1, 5, 37, 300) else (1,)):
                                g.set_small_path(mode)
                                D1, I1 = g.ivfpq_search(q, k, args)
                                s1 = g.last_stages(nq, P, max(R, k))
                                tag = (step, nq, has_rank, P, R, k, mode, cm)
                                assert D0.tobytes() == D1.tobytes() and np.array_equal(I0, I1), tag
                                for key in s0:
                                    assert s0[key].tobytes() == s1[key].tobytes(), (tag, key)
            if step == 0:
                (D, I, st), (Dg, Ig) = run_both(case, g, case["q"][:3], 10, 8, 100, metric, True)
                compare_exact(D, I, Dg, Ig)
    finally:
        g.close()


def test_large_batch_search_with_fused_coarse_matches_oracle():
    """A whole search at a size where the coarse quantizer runs without the distance matrix (csrc/coarse.hip:
    >= 4096 queries, >= 2048 lists): stage tables and results against the oracle (GEMM-form coarse, as faiss above
    20 queries).  Random codebooks stand in for training: parity does not care how good they are."""
    d, nlist, M, N, nq = 32, 2048, 8, 40000, 4200
    rng = np.random.default_rng(7)
    base = synth.sift_like(N, d=d, seed=11)
    q = synth.sift_like(nq, d=d, seed=12)
    cc = base[rng.choice(N, nlist, replace=False)].copy()
    pq = (rng.standard_normal((M, 256, d // M)) * 20).astype(np.float32)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2, bucket_init_size=100)
    o.set_trained(cc, pq, None)
    B.lib().go_set_assign_mode(0)
    assert o.add(base)
    o.set_raw(base)
    case = dict(d=d, nlist=nlist, M=M, N=N, nq=nq, metric=B.METRIC_L2, base=base, q=q, cc=cc, pq=pq, oracle=o)
    g = fixtures.load_hip(case, bucket_init_size=100)
    try:
        for has_rank in (True, False):
            (D, I, st), (Dg, Ig) = run_both(case, g, q, 10, 16, 100, B.METRIC_L2, has_rank, coarse_mode=1)
            sg = g.last_stages(nq, 16, 100)
            assert st["coarse_dis"].tobytes() == sg["coarse_dis"].tobytes()
            excluded = compare_search_exact(D, I, st, Dg, Ig, sg)
            assert excluded <= nq // 100
        g.set_coarse_fused(False)      # and the matrix path gives the same bytes
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=100, has_rank=True, coarse_mode=1, **WIDE)
        D0, I0 = g.ivfpq_search(q, 10, args)
        g.set_coarse_fused(True)
        D1, I1 = g.ivfpq_search(q, 10, args)
        assert D0.tobytes() == D1.tobytes() and np.array_equal(I0, I1)
    finally:
        g.close()


def test_multi_vector_documents(case):
    """Documents with several vectors (VIDMgr::VID2DocID, vector/raw_vector_common.h:90-95): the delete bitmap, the
    request's range bitmaps and the device columns are tested on the DOC id of a scanned vector
    (GammaSearchCondition::IsValid, common/gamma_common_data.h:99-108); labels stay vector ids."""
    N, q = case["N"], case["q"]
    rng = np.random.default_rng(77)
    per_doc = rng.integers(1, 4, size=N)                       # 1..3 vectors per document
    v2d = np.repeat(np.arange(N), per_doc)[:N].astype(np.int32)
    ndocs = int(v2d[-1]) + 1
    g = fixtures.load_hip(case)
    g.vid2docid_append(v2d[:N // 3])
    g.vid2docid_append(v2d[N // 3:])                           # grows
    assert g.vid2docid_count() == N
    price = rng.integers(0, 1000, size=ndocs).astype(np.int64)   # one value per DOC
    g.field_append(1, price)
    dead_docs = rng.choice(ndocs, ndocs // 6, replace=False)
    bm = np.zeros((ndocs >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, dead_docs >> 3, (1 << (dead_docs & 7)).astype(np.uint8))
    g.bitmap_upload(bm, ndocs)
    docs = rng.choice(ndocs, ndocs // 2, replace=False)
    try:
        o = case["oracle"]
        for has_rank in (True, False):
            ctx = B.make_ctx(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)], vid2docid=v2d, **WIDE)
            D, I, st = o.search(q, 10, 8, recall_num=100, has_rank=has_rank, metric=B.METRIC_L2, ctx=ctx, coarse_mode=0,
                                want_stages=True)
            args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=has_rank, coarse_mode=0,
                                  range_filters=[api.make_range_filter(docs)], **WIDE)
            Dg, Ig = g.ivfpq_search(q, 10, args)
            compare_exact(D, I, Dg, Ig)
            assert not np.isin(v2d[Ig[Ig >= 0]], dead_docs).any()
        # a device column clause on the doc's value
        mask = (price[v2d] >= 200) & (price[v2d] < 700)
        ctx = B.make_ctx(docids_bitmap=bm, range_filters=[B.make_range_filter(np.unique(v2d[mask]))], vid2docid=v2d, **WIDE)
        D, I = o.search(q, 10, 8, recall_num=100, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=0)
        Dg, Ig = g.ivfpq_search(q, 10, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, coarse_mode=0,
                                                      field_filters=[(1, 200, 700, True, False)], **WIDE))
        compare_exact(D, I, Dg, Ig)
        # flat search
        ctx = B.make_ctx(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)], vid2docid=v2d, **WIDE)
        Df, If = B.flat_search(case["base"], q[:8], 10, B.METRIC_L2, ctx)
        Dg, Ig = g.flat_search(q[:8], 10, api.SearchArgs(metric=api.METRIC_L2, range_filters=[api.make_range_filter(docs)],
                                                         **WIDE))
        compare_exact(Df, If, Dg, Ig)
    finally:
        g.close()


def test_small_batch_path_parameter_corners():
    """The small-batch chain at the edges of its gate (512 queries, 128 probes, recall_num 1024, k up to recall_num, one
    list, nprobe = nlist, empty probed lists) against the regular chain, byte for byte."""
    rng = np.random.default_rng(99)
    for (d, M, nlist, N, metric) in ((16, 4, 128, 6000, B.METRIC_L2), (24, 8, 1, 3000, B.METRIC_IP),
                                     (40, 8, 200, 1500, B.METRIC_L2)):   # 200 lists for 1500 vectors: many are empty
        case = fixtures.trained_case(d=d, nlist=nlist, M=M, N=N, nq=64, metric=metric)
        g = fixtures.load_hip(case)
        hip_metric = api.METRIC_L2 if metric == B.METRIC_L2 else api.METRIC_IP
        try:
            big = synth.sift_like(512, d=d, seed=777)
            for nq, P, R, k, has_rank in ((512, min(128, nlist), 1024, 1024, True), (512, min(128, nlist), 1000, 7, False),
                                          (1, nlist if nlist <= 128 else 128, 1024, 1024, True), (19, 1, 1, 1, True),
                                          (16, min(65, nlist), 64, 64, False), (17, min(127, nlist), 300, 300, True),
                                          (255, min(3, nlist), 5, 50, True)):
                q = big[:nq]
                for cm in (-1, 0, 1):
                    args = api.SearchArgs(metric=hip_metric, nprobe=P, recall_num=R, has_rank=has_rank, coarse_mode=cm, **WIDE)
                    g.set_small_path(0)
                    D0, I0 = g.ivfpq_search(q, k, args)
                    for mode in (1, 2):
                        g.set_small_path(mode)
                        D1, I1 = g.ivfpq_search(q, k, args)
                        assert D0.tobytes() == D1.tobytes() and np.array_equal(I0, I1), (d, nq, P, R, k, has_rank, cm, mode)
        finally:
            g.close()


class _env:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        import os
        self.old = os.environ.get(self.name)
        if self.value is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.value

    def __exit__(self, *a):
        import os
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old


@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
def test_large_filtered_batches_run_over_compacted_lists(metric):
    """Batches that test several times as many entries as the index holds have the lists cut down once per call to the
    entries that pass (delete bitmap, superseded slots of updated vectors, range / field clauses; kernels.hip
    k_compact_lists) and then run unfiltered: results byte for byte those of the per-code predicate, and the oracle's."""
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=metric)
    hip_metric = api.METRIC_L2 if metric == B.METRIC_L2 else api.METRIC_IP
    g = fixtures.load_hip(case)
    N = case["N"]
    rng = np.random.default_rng(11)
    try:
        dead = rng.choice(N, N // 6, replace=False)
        bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        g.bitmap_upload(bm, N)
        g.delete(dead)
        price = rng.integers(0, 1000, size=N).astype(np.int64)
        g.field_append(1, price)
        q = synth.sift_like(700, d=32, seed=4242)
        keep = rng.choice(N, N // 10, replace=False)
        for kw_g, kw_o in ((dict(), dict()),
                           (dict(range_filters=[api.make_range_filter(keep)]), dict(range_filters=[B.make_range_filter(keep)])),
                           (dict(field_filters=[(1, 100, 200, True, False)]),
                            dict(range_filters=[B.make_range_filter(np.nonzero((price >= 100) & (price < 200))[0])]))):
            for has_rank in (True, False):
                args = api.SearchArgs(metric=hip_metric, nprobe=16, recall_num=120, has_rank=has_rank, **WIDE, **kw_g)
                res = []
                for mode in ("0", "1", None):   # None: the library's estimate (700 x 16 x 312 >= 4 x 20000: compacted)
                    with _env("GAMMA_HIP_LIST_COMPACT", mode):
                        res.append(g.ivfpq_search(q, 10, args))
                for D1, I1 in res[1:]:
                    assert res[0][0].tobytes() == D1.tobytes() and np.array_equal(res[0][1], I1)
                ctx = B.make_ctx(docids_bitmap=bm, **WIDE, **kw_o)
                o = case["oracle"]   # (the shared fixture's oracle: validity comes from the context's bitmap only)
                Do, Io = o.search(q[:200], 10, 16, recall_num=120, has_rank=has_rank, metric=metric, ctx=ctx)[:2]
                compare_exact(Do, Io, res[1][0][:200], res[1][1][:200])
    finally:
        g.close()


def test_term_rows_can_be_rewritten(case, hip):
    """gamma_hip_term_update: a doc's items rewritten in place (fewer), at the end of the item array (more), to nothing;
    gamma_hip_field_update for the numeric column -- the filtered search follows."""
    g = hip
    N = len(case["base"])
    rng = np.random.default_rng(77)
    docs_items = [list(rng.choice(6, size=int(rng.integers(0, 4)), replace=False)) for _ in range(N)]
    vals = rng.integers(0, 100, size=N).astype(np.int64)
    g.term_append(31, docs_items)
    g.field_append(32, vals)
    q = case["q"][:16]

    def check():
        want = np.nonzero(np.array([2 in d for d in docs_items]) & (vals < 50))[0]
        a = api.SearchArgs(metric=api.METRIC_L2, term_filters=[(31, 1, [2])], field_filters=[(32, 0, 50, True, False)], **WIDE)
        b = api.SearchArgs(metric=api.METRIC_L2, range_filters=[api.make_range_filter(want)], **WIDE)
        Da, Ia = g.flat_search(q, 10, a)
        Db, Ib = g.flat_search(q, 10, b)
        assert Da.tobytes() == Db.tobytes() and np.array_equal(Ia, Ib)

    check()
    for doc in rng.choice(N, size=500, replace=False):
        docs_items[doc] = list(rng.choice(6, size=int(rng.integers(0, 6)), replace=False))
        g.term_update(31, int(doc), docs_items[doc])
        vals[doc] = int(rng.integers(0, 100))
        g.field_update(32, int(doc), vals[doc:doc + 1])
    check()


def test_shadow_lists_are_reused_until_a_writer_runs():
    """Large batches over an index with deleted documents run over shadow lists cut down to the live entries
    (compact_lists_for_call); without a clause of the call's own the shadow lists are kept for the next call and rebuilt
    only after a writer (Delete, Add, Update, bitmap) has run.  Every call == the oracle with the same deletes."""
    import os
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    o = case["oracle"]
    g = fixtures.load_hip(case)
    q = synth.sift_like(700, d=32, seed=5)
    N = case["N"]
    rng = np.random.default_rng(8)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    old = os.environ.get("GAMMA_HIP_LIST_COMPACT")
    os.environ["GAMMA_HIP_LIST_COMPACT"] = "1"
    try:
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True, **WIDE)
        for step in range(4):
            dead = rng.choice(N, 800, replace=False)
            np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
            g.bitmap_upload(bm, N)
            ctx = B.make_ctx(docids_bitmap=bm, **WIDE)
            D, I = o.search(q, 10, 8, recall_num=100, has_rank=True, metric=B.METRIC_L2, ctx=ctx)
            for rep in range(3):        # the first call builds the shadow lists, the others reuse them
                Dg, Ig = g.ivfpq_search(q, 10, args)
                compare_exact(D, I, Dg, Ig)
            # a call with a clause of its own in between must not leave ITS shadow lists behind
            docs = rng.choice(N, N // 2, replace=False)
            a2 = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True,
                                range_filters=[api.make_range_filter(docs)], **WIDE)
            D2, I2 = o.search(q, 10, 8, recall_num=100, has_rank=True, metric=B.METRIC_L2,
                              ctx=B.make_ctx(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)], **WIDE))
            Dg2, Ig2 = g.ivfpq_search(q, 10, a2)
            compare_exact(D2, I2, Dg2, Ig2)
            Dg, Ig = g.ivfpq_search(q, 10, args)
            compare_exact(D, I, Dg, Ig)
    finally:
        if old is None:
            os.environ.pop("GAMMA_HIP_LIST_COMPACT", None)
        else:
            os.environ["GAMMA_HIP_LIST_COMPACT"] = old
        g.close()


def test_profile_levels_and_the_state_the_pair_offset_kernel_clears(case):
    """gamma_hip_profile_enable: 1 = an event pair around every stage + the scanned-code counter, 2 = around the scan launch
    alone (what a throughput measurement that wants the scan's duration live can afford), 0 = none.  The results do not
    depend on the level, and neither on what the previous call left in the per-call state that k_pair_offsets clears
    (tie flags, repair list, ready words, query-order histogram): calls of different sizes alternate on one handle."""
    g = fixtures.load_hip(case)
    try:
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=60, has_rank=True, coarse_mode=1, **WIDE)
        qa = synth.sift_like(9000, d=case["d"], seed=77)    # > 8192 queries: the grid-wide query order
        qb = synth.sift_like(700, d=case["d"], seed=78)
        ref = {}
        for name, q in (("a", qa), ("b", qb)):
            ref[name] = g.ivfpq_search(q, 10, args)
        for level in (1, 2, 0, 2, 1):
            g.profile_enable(level)
            g.profile_reset()
            for name, q in (("a", qa), ("b", qb), ("b", qb), ("a", qa)):
                D, I = g.ivfpq_search(q, 10, args)
                assert D.tobytes() == ref[name][0].tobytes() and np.array_equal(I, ref[name][1]), (level, name)
            prof = g.profile()
            launches = {n: prof[n][1] for n in ("coarse", "tables", "scan", "select", "rerank")}
            if level == 0:
                assert not any(launches.values()) and prof["scan_bytes"] == 0
            elif level == 2:
                assert launches["scan"] == 4 and prof["scan"][0] > 0.0
                assert not any(v for n, v in launches.items() if n != "scan") and prof["scan_bytes"] == 0
            else:
                assert all(v >= 4 for v in launches.values()), launches
                assert prof["scan_bytes"] > 0
        g.profile_enable(0)
    finally:
        g.close()


def test_large_host_buffer_calls_are_the_device_pointer_call(c3):
    """A large host-buffer call (what RetrievalModel::Search hands over) == the device-pointer call, bit for bit: several
    sizes, with and without re-rank, ties on and off, several scan chunks (a small workspace budget).  (Round 4 built a
    pipelined form of this call -- queries uploaded in chunks behind which the coarse quantizer ran, results fetched beside
    the tie replay -- and measured it SLOWER than upload / search / download: tools/host_call_bench.py, DESIGN.md.)"""
    import torch
    g = c3["g"]
    q = synth.sift_like(16384 + 777, d=128, seed=4321)
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(q).to(dev)
    try:
        for nq, k, R, has_rank, ties in ((16384, 10, 200, True, 0), (4096, 10, 200, True, 0), (16384 + 777, 10, 200, True, 0),
                                         (5000, 10, 100, False, 0), (8192, 50, 200, True, -1), (4100, 1, 100, True, 0)):
            args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=R, has_rank=has_rank, min_score=0.0, max_score=1e30,
                                  exact_ties=ties)
            D = torch.empty((nq, k), dtype=torch.float32, device=dev)
            I = torch.empty((nq, k), dtype=torch.int64, device=dev)
            g.ivfpq_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
            g.synchronize()
            g.tie_stats(reset=True)
            Dh, Ih = g.ivfpq_search(q[:nq], k, args)
            assert Dh.tobytes() == D.cpu().numpy().tobytes() and np.array_equal(Ih, I.cpu().numpy()), (nq, k, R, has_rank, ties)
            if ties == 0 and has_rank and nq >= 8192:
                assert g.tie_stats()["replayed"] > 0       # (integer data: some rows did go through the replay and the patch)
        g.set_dist_budget(64 << 20)      # several scan chunks per call
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True, min_score=0.0, max_score=1e30)
        nq, k = 16384, 10
        D = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        g.ivfpq_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
        g.synchronize()
        Dh, Ih = g.ivfpq_search(q[:nq], k, args)
        assert Dh.tobytes() == D.cpu().numpy().tobytes() and np.array_equal(Ih, I.cpu().numpy())
    finally:
        g.set_dist_budget(32 << 30)
