"""CPU: the oracle (oracle/gamma_oracle.c) against golden vectors generated from the real
faiss 1.7.1 / the reference's realtime sources (tests/gen_golden.py).  Bit-exact."""
import os

import numpy as np
import pytest

from gamma_amd import synth
from oracle import binding as B
from tests.parity import compare_topk

G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_primitives():
    z = np.load(os.path.join(G, "prims.npz"))
    L = B.lib()
    for d in z["dims"]:
        x, y, res = z["x_%d" % d], z["y_%d" % d], z["res_%d" % d]
        got = np.zeros_like(res)
        for i in range(x.shape[0]):
            got[i, 0] = L.go_fvec_L2sqr(B._fp(x[i]), B._fp(y[i]), int(d))
            got[i, 1] = L.go_fvec_inner_product(B._fp(x[i]), B._fp(y[i]), int(d))
            got[i, 2] = L.go_fvec_norm_L2sqr(B._fp(x[i]), int(d))
        assert np.array_equal(bits(got), bits(res)), "d=%d" % d
    for d in z["ny_dims"]:
        x, y = z["nyx_%d" % d], z["nyy_%d" % d]
        ip, l2 = np.empty(len(y), np.float32), np.empty(len(y), np.float32)
        L.go_fvec_inner_products_ny(B._fp(ip), B._fp(x), B._fp(y), int(d), len(y))
        L.go_fvec_L2sqr_ny(B._fp(l2), B._fp(x), B._fp(y), int(d), len(y))
        assert np.array_equal(bits(ip), bits(z["nyip_%d" % d])), "ip_ny d=%d" % d
        assert np.array_equal(bits(l2), bits(z["nyl2_%d" % d])), "l2_ny d=%d" % d
    c = np.empty(512, np.float32)
    L.go_fvec_madd(512, B._fp(z["madd_a"]), -2.0, B._fp(z["madd_b"]), B._fp(c))
    assert np.array_equal(bits(c), bits(z["madd_c"]))


def test_heap_mechanics_with_ties():
    z = np.load(os.path.join(G, "heap.npz"))
    L = B.lib()
    for ci, (ks, k, n) in enumerate(z["cases"]):
        vals = np.ascontiguousarray(z["vals_%d" % ci])
        ids = np.arange(n, dtype=np.int64)
        hv, hi = np.empty(k, np.float32), np.empty(k, np.int64)
        sv, si = np.empty(k, np.float32), np.empty(k, np.int64)
        pv, pi = np.empty(k, np.float32), np.empty(k, np.int64)
        L.go_heap_stream(int(ks), int(k), int(n), B._fp(vals), B._ip(ids), B._fp(hv), B._ip(hi),
                         B._fp(sv), B._ip(si))
        L.go_heap_pop_push_stream(int(ks), int(k), int(n), B._fp(vals), B._ip(ids), B._fp(pv), B._ip(pi))
        for nm, arr in (("hv", hv), ("hi", hi), ("sv", sv), ("si", si), ("pv", pv), ("pi", pi)):
            assert arr.tobytes() == z["%s_%d" % (nm, ci)].tobytes(), (nm, ks, k, n)


def test_reservoir_mechanics_with_ties():
    """tests/golden/reservoir_ties.npz: streams of tied keys through the compiled library's ReservoirTopN (what
    knn_L2sqr / knn_inner_product collect through from k = 100 on) -- the restatement leaves the same labels at every
    rank, and the result heap would NOT (the fixture tells the two apart)."""
    z = np.load(os.path.join(G, "reservoir_ties.npz"))
    L = B.lib()
    heap_differs = 0
    for ci, (ks, k, n, hi) in enumerate(z["cases"]):
        keys = np.ascontiguousarray(z["keys_%d" % ci])
        sv, si = np.empty(k, np.float32), np.empty(k, np.int64)
        L.go_reservoir_stream(int(ks), int(k), int(n), B._fp(keys), None, B._fp(sv), B._ip(si))
        assert sv.tobytes() == z["D_%d" % ci].tobytes() and np.array_equal(si, z["I_%d" % ci]), (ks, k, n, hi)
        hv, hi_ = np.empty(k, np.float32), np.empty(k, np.int64)
        tv, ti = np.empty(k, np.float32), np.empty(k, np.int64)
        L.go_heap_stream(int(ks), int(k), int(n), B._fp(keys), B._ip(np.arange(n, dtype=np.int64)), B._fp(hv), B._ip(hi_),
                         B._fp(tv), B._ip(ti))
        assert tv.tobytes() == sv.tobytes()          # same keys
        heap_differs += int(not np.array_equal(ti, si))
    assert heap_differs > len(z["cases"]) // 2


def load_ivfpq(name):
    z = np.load(os.path.join(G, name + ".npz"))
    d, nlist, M, N = int(z["d"]), int(z["nlist"]), int(z["M"]), int(z["N"])
    base = synth.sift_like(N, d=d, seed=1234)
    if int(z["normalize"]):
        base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    assert base.astype(np.float64).sum() == z["base_sum"][0], "synthetic generator drifted"
    return z, base


@pytest.mark.parametrize("name", ["ivfpq_l2_d32", "ivfpq_l2_d64", "ivfpq_ip_d48"])
def test_ivfpq_pipeline(name):
    z, base = load_ivfpq(name)
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    metric, nprobe, R = int(z["metric"]), int(z["nprobe"]), int(z["R"])
    o = B.OracleIVFPQ(d, nlist, M, 8, metric)
    o.set_trained(z["cc"], z["pq"], None)
    # precomputed table (faiss:IndexIVFPQ.cpp:453-479)
    assert o.table().tobytes() == z["table"].tobytes()
    # Add path arithmetic (gamma_index_ivfpq.cc:455-472), exact assign
    B.lib().go_set_assign_mode(0)
    lno, codes = o.encode(base[:500])
    assert np.array_equal(lno, z["enc_lno"]) and np.array_equal(codes, z["enc_codes"])
    assert o.add(base)
    off = 0
    for l in range(nlist):
        n = int(z["list_sizes"][l])
        ids, cds = o.get_list(l)
        assert np.array_equal(ids, z["list_ids"][off:off + n])
        assert np.array_equal(cds, z["list_codes"][off:off + n])
        off += n
    o.set_raw(base)
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    for m, tag in ((B.METRIC_L2, "l2"), (B.METRIC_IP, "ip")):
        D, I, st = o.search(z["q"], 5, nprobe, recall_num=R, has_rank=False, metric=m, ctx=ctx,
                            coarse_mode=0, want_stages=True)
        assert st["coarse_dis"].tobytes() == z["coarse_dis"].tobytes()
        assert np.array_equal(st["coarse_idx"], z["coarse_idx"])
        # R-stage == faiss::IndexIVFPQ::search(k=R): identical, including the heap's tie order
        assert st["recall_dis"].tobytes() == z["rdis_" + tag].tobytes()
        assert np.array_equal(st["recall_ids"], z["rids_" + tag])
        # has_rank=false output = first k of the sorted R-heap (gamma_index_ivfpq.cc:681-696)
        assert D.tobytes() == z["rdis_" + tag][:, :5].tobytes()
        assert np.array_equal(I, z["rids_" + tag][:, :5])


@pytest.mark.parametrize("name", ["ivfpq_l2_mode0_d32", "ivfpq_l2_mode0_d64", "ivfpq_l2_mode0_d96", "ivfpq_l2_mode0_d128m8"])
def test_ivfpq_table_mode_0_pipeline(name):
    """L2 table mode 0 (faiss:IndexIVFPQ.cpp:441-449: the table would exceed precomputed_table_max_bytes, none is built;
    index/impl/gamma_index_ivfpq.h:239-245: residual q - centroid, compute_distance_table on it, dis0 = 0).  The fixtures
    come from the compiled library with its extern limit lowered (tests/gen_golden.py mode0)."""
    z, base = load_ivfpq(name)
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    nprobe, R = int(z["nprobe"]), int(z["R"])
    assert int(z["table_mode"]) == 0 and z["table"].size == 0
    L = B.lib()
    saved = L.go_get_precomputed_table_max_bytes()
    L.go_set_precomputed_table_max_bytes(int(z["table_max_bytes"]))
    try:
        o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
        o.set_trained(z["cc"], z["pq"], None)
        assert o.use_precomputed_table() == 0 and o.table() is None
        # one byte more and the table is built: the rule is `>` (faiss:IndexIVFPQ.cpp:442)
        L.go_set_precomputed_table_max_bytes(nlist * M * 1024)
        o1 = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
        o1.set_trained(z["cc"], z["pq"], None)
        assert o1.use_precomputed_table() == 1 and o1.table() is not None
    finally:
        L.go_set_precomputed_table_max_bytes(saved)
    L.go_set_assign_mode(0)
    assert o.add(base)
    off = 0
    for l in range(nlist):
        n = int(z["list_sizes"][l])
        ids, cds = o.get_list(l)
        assert np.array_equal(ids, z["list_ids"][off:off + n])
        assert np.array_equal(cds, z["list_codes"][off:off + n])
        off += n
    o.set_raw(base)
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    differs_from_mode1 = 0
    for m, tag in ((B.METRIC_L2, "l2"), (B.METRIC_IP, "ip")):
        D, I, st = o.search(z["q"], 5, nprobe, recall_num=R, has_rank=False, metric=m, ctx=ctx,
                            coarse_mode=0, want_stages=True)
        assert np.array_equal(st["coarse_idx"], z["coarse_idx"])
        assert st["recall_dis"].tobytes() == z["rdis_" + tag].tobytes()
        assert np.array_equal(st["recall_ids"], z["rids_" + tag])
        assert D.tobytes() == z["rdis_" + tag][:, :5].tobytes() and np.array_equal(I, z["rids_" + tag][:, :5])
        if m == B.METRIC_L2:   # the branch is not a no-op: table mode 1 rounds differently
            o1.add(base)
            o1.set_raw(base)
            _, _, st1 = o1.search(z["q"], 5, nprobe, recall_num=R, has_rank=False, metric=m, ctx=ctx,
                                  coarse_mode=0, want_stages=True)
            differs_from_mode1 = int((st1["recall_dis"].view(np.uint32) != st["recall_dis"].view(np.uint32)).sum())
    assert differs_from_mode1 > 0


def test_gemm_form_coarse_agrees_with_exact():
    """The 'BLAS form' coarse distances (restated GEMM) differ from the exact ones only by
    rounding: same probe sets except near-ties, distances within 1e-4 relative."""
    z, base = load_ivfpq("ivfpq_l2_d64")
    D0, I0 = B.knn_L2sqr(z["q"], z["cc"], int(z["nprobe"]), mode=0)
    D1, I1 = B.knn_L2sqr(z["q"], z["cc"], int(z["nprobe"]), mode=1)
    assert np.allclose(D0, D1, rtol=1e-4, atol=1e-3)
    assert (I0 == I1).mean() > 0.98


def test_realtime_lists_replay():
    z = np.load(os.path.join(G, "realtime.npz"))
    nlist, cs = int(z["nlist"]), int(z["cs"])
    o = B.OracleIVFPQ(cs * 1, nlist, cs, 8, B.METRIC_L2, int(z["binit"]), int(z["bmax"]))
    L = B.lib()
    bm = np.zeros(int(z["nbits"]) // 8 + 1, dtype=np.uint8)
    o.set_docids_bitmap(bm)
    next_vid, snap = 0, 0
    snap_after = set(int(x) for x in z["snap_after"])
    for i in range(int(z["nops"])):
        op, a, b, ok = [int(x) for x in z["op_%d" % i]]
        pl = z["pl_%d" % i]
        if op == 0:
            keys = np.arange(next_vid, next_vid + b, dtype=np.int64)
            got = o.add_keys(a, keys, np.ascontiguousarray(pl))
            assert int(got) == ok
            if ok:
                next_vid += b
        elif op == 1:
            # RealTimeMemData::Update with an explicit target list and code
            _update_to(o, L, a, b, np.ascontiguousarray(pl))
        elif op == 2:
            for v in pl:
                bm[int(v) >> 3] |= np.uint8(1 << (int(v) & 7))
            o.delete(pl.astype(np.int64))
        elif op == 3:
            o.compact_if_need()
        if i in snap_after:
            for l in range(nlist):
                ids, codes = o.get_list(l)
                assert np.array_equal(ids, z["ids_%d_%d" % (snap, l)]), (snap, l)
                assert np.array_equal(codes, z["codes_%d_%d" % (snap, l)]), (snap, l)
            caps = np.array([o.list_capacity(l) for l in range(nlist)])
            assert np.array_equal(caps, z["caps_%d" % snap]), snap
            vp = np.array([o.vid_pos(v) for v in range(len(z["vp_%d" % snap]))])
            assert np.array_equal(vp, z["vp_%d" % snap]), snap
            snap += 1
    assert snap == int(z["nsnaps"])


def _update_to(o, L, list_no, vid, code):
    rc = L.go_ivfpq_update_code(o.h, list_no, vid, B._up(code))
    assert rc >= 0


def test_flat_matches_knn():
    rng = np.random.default_rng(3)
    base = rng.integers(0, 50, size=(500, 24)).astype(np.float32)   # many ties
    q = rng.integers(0, 50, size=(7, 24)).astype(np.float32)
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    D, I = B.flat_search(base, q, 10, B.METRIC_L2, ctx)
    Dk, Ik = B.knn_L2sqr(q, base, 10, mode=0)
    # same distances; heap_pop+heap_push vs heap_replace_top only reorder exact ties
    compare_topk(Dk, Ik, D, I)


def load_ties(tag, name="ivfpq_ties_d32"):
    """tests/golden/ivfpq_ties_d32.npz, ivfpq_ties_c4shape.npz (gen_golden.gen_ivfpq_ties): duplicated integer base
    vectors, every expected table built with the real faiss primitives and heaps.  The second file has the cuts of
    the C4 configuration: more than 4096 lists, 64 probes, recall_num 100."""
    z = np.load(os.path.join(G, name + ".npz"))
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    b0 = synth.sift_like(int(z["N0"]), d=d, seed=1234)
    base = np.ascontiguousarray(b0[z["pick"]])
    metric = B.METRIC_L2 if tag == "l2" else B.METRIC_IP
    o = B.OracleIVFPQ(d, nlist, M, 8, metric)
    o.set_trained(z["cc_" + tag], z["pq_" + tag], None)
    sizes = z["list_sizes_" + tag]
    off = 0
    for l in range(nlist):
        n = int(sizes[l])
        if n:
            o.add_keys(l, z["list_ids_" + tag][off:off + n], z["list_codes_" + tag][off:off + n])
        off += n
    o.set_raw(base)
    return z, o, base, metric


@pytest.mark.parametrize("name", ["ivfpq_ties_d32", "ivfpq_ties_c4shape"])
@pytest.mark.parametrize("tag", ["l2", "ip"])
def test_tie_heavy_golden_labels_and_ranks_exact(tag, name):
    """Equal ADC distances straddle the recall_num cut in most queries, equal exact distances the k cut: the
    oracle's heaps must leave exactly what the real library's heaps leave -- labels at every rank."""
    from tests.parity import compare_exact
    z, o, base, metric = load_ties(tag, name)
    nprobe, R, k = int(z["nprobe"]), int(z["R"]), int(z["k"])
    assert int(z["ncut_" + tag][0]) > 10
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    for has_rank, nm in ((True, "rank"), (False, "norank")):
        D, I, st = o.search(z["q"], k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx,
                            coarse_mode=0, want_stages=True)
        assert st["coarse_dis"].tobytes() == z["coarse_dis_" + tag].tobytes()
        assert np.array_equal(st["coarse_idx"], z["coarse_idx_" + tag])
        compare_exact(z["rdis_" + tag], z["rids_" + tag], st["recall_dis"], st["recall_ids"])
        compare_exact(z["D_%s_%s" % (nm, tag)], z["I_%s_%s" % (nm, tag)], D, I)


def load_blas_c3shape():
    """tests/golden/ivfpq_blas_c3shape.npz (tests/gen_golden.py gen_blas_coarse): a C3-shaped index trained, filled and
    searched by the compiled library at its DEFAULT blas threshold.  Base and queries come from the portable generator."""
    z = np.load(os.path.join(G, "ivfpq_blas_c3shape.npz"))
    N, d, nq = int(z["N"]), int(z["d"]), int(z["nq"])
    base = synth.sift_like(N, d=d, seed=1234)
    q = synth.sift_like(nq, d=d, seed=4321)
    assert float(base.astype(np.float64).sum()) == float(z["base_sum"][0]) and float(q.astype(np.float64).sum()) == float(z["q_sum"][0])
    return z, base, q


def check_blas_c3shape_lists(z, get_list):
    """the lists an Add path built == the library's: sizes and checksums of ids and codes per list"""
    M = int(z["M"])
    w = 1 + np.arange(M)
    for l in range(int(z["nlist"])):
        ids, codes = get_list(l)
        assert len(ids) == int(z["list_sizes"][l]), l
        assert int(ids.sum()) == int(z["list_idsum"][l]) and int((codes.astype(np.int64) * w).sum()) == int(z["list_codesum"][l]), l


def test_default_blas_path_of_the_library_at_the_c3_shape():
    """The oracle's DEFAULT path (coarse_mode -1: GEMM form from 20 queries on, Add with the library's assign rule) on a
    2048-query batch == the compiled library with its default BLAS threshold: lists, coarse assignment (distance bits and
    list order), final labels and distances -- all strictly.  "Bit-exact vs the faiss-CPU path" therefore holds for
    batched calls as the library actually runs them, not only with its BLAS switch disabled."""
    z, base, q = load_blas_c3shape()
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
    o.set_trained(z["cc"], z["pq"], None)
    B.lib().go_set_assign_mode(-1)
    try:
        for i0 in range(0, len(base), 50000):
            assert o.add(base[i0:i0 + 50000])
    finally:
        B.lib().go_set_assign_mode(0)
    check_blas_c3shape_lists(z, o.get_list)
    o.set_raw(base)
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    nprobe, R, k = int(z["nprobe"]), int(z["R"]), int(z["k"])
    D, I, st = o.search(q, k, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=-1, want_stages=True)
    assert st["coarse_dis"].tobytes() == z["coarse_dis_blas"].tobytes()
    assert np.array_equal(st["coarse_idx"], z["coarse_idx_blas"].astype(np.int64))
    assert D.tobytes() == z["D_blas"].tobytes() and np.array_equal(I, z["I_blas"].astype(np.int64))
    # the switch disabled (what the other goldens pin): a different, equally exact, answer
    D0, I0, st0 = o.search(q, k, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=0, want_stages=True)
    assert st0["coarse_dis"].tobytes() == z["coarse_dis_exact"].tobytes()
    assert np.array_equal(st0["coarse_idx"], z["coarse_idx_exact"].astype(np.int64))
    assert D0.tobytes() == z["D_exact"].tobytes() and np.array_equal(I0, z["I_exact"].astype(np.int64))
    assert not np.array_equal(z["coarse_idx_blas"], z["coarse_idx_exact"])
