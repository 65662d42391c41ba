#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference arithmetic: faiss 1.7.1 built from the
reference's vendored tarball and the reference's own realtime/ sources (oracle/_ref, built by
oracle/Makefile.ref -- only possible in the container that has /root/reference).

The fixtures are data only (inputs + expected outputs); the oracle, and on the GPU box the HIP
path, are checked against them.  Re-run:  python tests/gen_golden.py
Determinism: faiss::distance_compute_blas_threshold is raised so the coarse quantizer takes the
exact per-pair path (no BLAS summation order), OMP threads do not affect results.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")

from gamma_amd import synth  # noqa: E402
from oracle import binding as B  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_prims(R):
    rng = np.random.default_rng(7)
    out = {}
    dims = [1, 2, 3, 4, 5, 7, 8, 12, 15, 16, 17, 31, 32, 33, 64, 100, 128, 129, 768]
    out["dims"] = np.array(dims)
    for d in dims:
        x = (rng.standard_normal((6, d)) * rng.uniform(0.1, 100, size=(6, 1))).astype(np.float32)
        y = (rng.standard_normal((6, d)) * 3).astype(np.float32)
        res = np.zeros((6, 3), dtype=np.float32)
        for i in range(6):
            res[i, 0] = R.ref_fvec_L2sqr(B._fp(x[i]), B._fp(y[i]), d)
            res[i, 1] = R.ref_fvec_inner_product(B._fp(x[i]), B._fp(y[i]), d)
            res[i, 2] = R.ref_fvec_norm_L2sqr(B._fp(x[i]), d)
        out["x_%d" % d], out["y_%d" % d], out["res_%d" % d] = x, y, res
    nyd = [1, 2, 4, 8, 12, 3, 6, 16]
    out["ny_dims"] = np.array(nyd)
    for d in nyd:
        ny = 256
        x = rng.standard_normal(d).astype(np.float32)
        y = rng.standard_normal((ny, d)).astype(np.float32)
        ip = np.empty(ny, np.float32)
        l2 = np.empty(ny, np.float32)
        R.ref_fvec_inner_products_ny(B._fp(ip), B._fp(x), B._fp(y), d, ny)
        R.ref_fvec_L2sqr_ny(B._fp(l2), B._fp(x), B._fp(y), d, ny)
        out["nyx_%d" % d], out["nyy_%d" % d], out["nyip_%d" % d], out["nyl2_%d" % d] = x, y, ip, l2
    a = rng.standard_normal(512).astype(np.float32)
    b = rng.standard_normal(512).astype(np.float32)
    c = np.empty(512, np.float32)
    R.ref_fvec_madd(512, B._fp(a), -2.0, B._fp(b), B._fp(c))
    out["madd_a"], out["madd_b"], out["madd_c"] = a, b, c
    np.savez_compressed(os.path.join(OUT, "prims.npz"), **out)


def gen_heap(R):
    rng = np.random.default_rng(11)
    out = {}
    cases = []
    for ks in (0, 1):
        for k in (1, 3, 10, 100):
            for n in (0, 5, 300):
                cases.append((ks, k, n))
    out["cases"] = np.array(cases)
    for ci, (ks, k, n) in enumerate(cases):
        vals = rng.integers(0, 25, size=n).astype(np.float32)  # many exact ties
        ids = np.arange(n, dtype=np.int64)
        hv, hi = np.empty(k, np.float32), np.empty(k, np.int64)
        sv, si = np.empty(k, np.float32), np.empty(k, np.int64)
        pv, pi = np.empty(k, np.float32), np.empty(k, np.int64)
        R.ref_heap_stream(ks, k, n, B._fp(vals), B._ip(ids), B._fp(hv), B._ip(hi), B._fp(sv), B._ip(si))
        R.ref_heap_pop_push_stream(ks, k, n, B._fp(vals), B._ip(ids), B._fp(pv), B._ip(pi))
        out["vals_%d" % ci] = vals
        for nm, arr in (("hv", hv), ("hi", hi), ("sv", sv), ("si", si), ("pv", pv), ("pi", pi)):
            out["%s_%d" % (nm, ci)] = arr
    np.savez_compressed(os.path.join(OUT, "heap.npz"), **out)


def gen_reservoir(R):
    """Streams of tied keys through the compiled library's ReservoirTopN: IndexFlatL2 / IndexFlatIP over one-dimensional
    integer rows, one query -- the distance of row j is vals[j] exactly (L2: (0 - v)^2 with v = sqrt-free small integers
    squared by the library itself; IP: 1 * v), k >= 100 takes the reservoir (faiss:utils/distances.cpp:307-358)."""
    rng = np.random.default_rng(23)
    out = {}
    cases = []
    R.ref_set_blas_threshold(1 << 30)
    for ks in (1, 0):
        for k in (100, 128, 200, 256):
            for n, hi in ((k, 3), (300, 2), (1000, 6), (4096, 12), (5000, 40)):
                if n >= k:
                    cases.append((ks, k, n, hi))
    out["cases"] = np.array(cases)
    for ci, (ks, k, n, hi) in enumerate(cases):
        v = rng.integers(0, hi, size=(n, 1)).astype(np.float32)
        D = np.empty((1, k), np.float32)
        I = np.empty((1, k), np.int64)
        if ks:
            x = np.zeros((1, 1), np.float32)
            R.ref_flat_l2_search(1, n, B._fp(v), 1, B._fp(x), k, B._fp(D), B._ip(I))
            keys = (v * v).ravel()
        else:
            x = np.ones((1, 1), np.float32)
            R.ref_flat_ip_search(1, n, B._fp(v), 1, B._fp(x), k, B._fp(D), B._ip(I))
            keys = v.ravel().copy()
        out["keys_%d" % ci] = keys
        out["D_%d" % ci] = D[0]
        out["I_%d" % ci] = I[0]
    np.savez_compressed(os.path.join(OUT, "reservoir_ties.npz"), **out)


def gen_ivfpq(R, name, d, nlist, M, N, nq, metric, nprobe, Rk, normalize=False, table_max_bytes=None):
    """table_max_bytes: the library's extern faiss::precomputed_table_max_bytes lowered for this index (restored
    afterwards) -- below nlist * M * 1 KiB the trained index stays in table mode 0 (faiss:IndexIVFPQ.cpp:441-449) and
    the L2 search scores with per-(query, list) residual tables (index/impl/gamma_index_ivfpq.h:239-245)."""
    saved = None
    if table_max_bytes is not None:
        saved = R.ref_get_precomputed_table_max_bytes()
        R.ref_set_precomputed_table_max_bytes(table_max_bytes)
    try:
        _gen_ivfpq(R, name, d, nlist, M, N, nq, metric, nprobe, Rk, normalize, table_max_bytes)
    finally:
        if saved is not None:
            R.ref_set_precomputed_table_max_bytes(saved)


def _gen_ivfpq(R, name, d, nlist, M, N, nq, metric, nprobe, Rk, normalize, table_max_bytes):
    base = synth.sift_like(N, d=d, seed=1234)
    q = synth.sift_like(nq, d=d, seed=4321)
    if normalize:
        base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
        q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    r = B.RefIVFPQ(d, nlist, M, 8, metric)
    r.train(base[:min(N, nlist * 64)])
    r.add(base)
    mode = r.use_precomputed_table()
    assert mode == (1 if table_max_bytes is None or nlist * M * 1024 <= table_max_bytes else 0)
    out = dict(d=d, nlist=nlist, M=M, N=N, nq=nq, metric=metric, nprobe=nprobe, R=Rk,
               normalize=int(normalize), q=q, table_mode=mode,
               table_max_bytes=-1 if table_max_bytes is None else table_max_bytes,
               cc=r.coarse_centroids(), pq=r.pq_centroids(), table=r.precomputed_table())
    # base is regenerated from the portable generator (not stored): only its checksum
    out["base_sum"] = np.array([base.astype(np.float64).sum()])
    sizes, ids, codes = [], [], []
    for l in range(nlist):
        i, c = r.get_list(l)
        sizes.append(len(i))
        ids.append(i)
        codes.append(c)
    out["list_sizes"] = np.array(sizes, dtype=np.int64)
    out["list_ids"] = np.concatenate(ids)
    out["list_codes"] = np.concatenate(codes)
    lno, cds = r.encode(base[:500])
    out["enc_lno"], out["enc_codes"] = lno, cds
    cd, ci = r.coarse(q, nprobe)
    out["coarse_dis"], out["coarse_idx"] = cd, ci
    for m, tag in ((B.METRIC_L2, "l2"), (B.METRIC_IP, "ip")):
        r.set_metric(m)
        D, I = r.search(q, Rk, nprobe)
        out["rdis_" + tag], out["rids_" + tag] = D, I
    r.set_metric(metric)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def gen_mode0(R):
    """L2 table mode 0: the table of each shape would be nlist * M KiB; the limit sits just below it."""
    gen_ivfpq(R, "ivfpq_l2_mode0_d32", 32, 32, 8, 6000, 40, B.METRIC_L2, 6, 64, table_max_bytes=32 * 8 * 1024 - 1)    # dsub 4
    gen_ivfpq(R, "ivfpq_l2_mode0_d64", 64, 32, 8, 6000, 40, B.METRIC_L2, 8, 100, table_max_bytes=1 << 16)             # dsub 8
    gen_ivfpq(R, "ivfpq_l2_mode0_d96", 96, 16, 8, 4000, 30, B.METRIC_L2, 4, 50, table_max_bytes=0)                    # dsub 12
    gen_ivfpq(R, "ivfpq_l2_mode0_d128m8", 128, 16, 8, 4000, 30, B.METRIC_L2, 4, 50, table_max_bytes=4096)             # dsub 16: the AVX fvec_L2sqr row


def gen_ivfpq_ties(R, name, d, nlist, M, N0, N, nq, nprobe, Rk, k):
    """Tie-heavy case: every base vector occurs ~N/N0 times (integer data), so equal codes share a list and
    equal ADC distances straddle every cut; exact distances tie as well.  Expected outputs, all from the real
    library: the recall-stage table (faiss IndexIVFPQ::search with k = recall_num) and the FINAL results of
    compute_dis for has_rank on / off, rebuilt here step by step -- candidate stream in scan order (real
    inner-product table, real fvec_madd, float32 adds in code order), real heap_replace_top stream
    (ref_heap_stream, which also returns the heap's ARRAY), real fvec_L2sqr / fvec_inner_product on the array
    order, real heap_pop + heap_push stream + heap_reorder (index/impl/gamma_index_ivfpq.cc:642-697)."""
    rng = np.random.default_rng(23)
    b0 = synth.sift_like(N0, d=d, seed=1234)
    pick = rng.integers(0, N0, size=N).astype(np.int32)
    base = np.ascontiguousarray(b0[pick])
    q = synth.sift_like(nq, d=d, seed=4321)
    q[: nq // 4] = b0[rng.integers(0, N0, size=nq // 4)]          # some queries ARE base vectors
    out = dict(d=d, nlist=nlist, M=M, N0=N0, N=N, nq=nq, nprobe=nprobe, R=Rk, k=k, pick=pick, q=q)
    for metric, tag in ((B.METRIC_L2, "l2"), (B.METRIC_IP, "ip")):
        r = B.RefIVFPQ(d, nlist, M, 8, metric)
        r.train(b0)
        r.add(base)
        assert r.use_precomputed_table() == 1 or metric == B.METRIC_IP
        cc, pq = r.coarse_centroids(), r.pq_centroids()
        out["cc_" + tag], out["pq_" + tag] = cc, pq
        sizes, ids, codes = [], [], []
        for l in range(nlist):
            i, c = r.get_list(l)
            sizes.append(len(i)); ids.append(i); codes.append(c)
        out["list_sizes_" + tag] = np.array(sizes, dtype=np.int64)
        out["list_ids_" + tag] = np.concatenate(ids)
        out["list_codes_" + tag] = np.concatenate(codes)
        cd, ci = r.coarse(q, nprobe)
        out["coarse_dis_" + tag], out["coarse_idx_" + tag] = cd, ci
        Dr, Ir = r.search(q, Rk, nprobe)
        out["rdis_" + tag], out["rids_" + tag] = Dr, Ir
        table = r.precomputed_table() if metric == B.METRIC_L2 else None
        ks = 1 if metric == B.METRIC_L2 else 0
        tsz = M * 256
        fin = {True: (np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)),
               False: (np.empty((nq, k), np.float32), np.empty((nq, k), np.int64))}
        ntie_cut = 0
        for qi in range(nq):
            xq = np.ascontiguousarray(q[qi])
            st2 = r.inner_prod_table(xq).reshape(-1)
            sv, si = [], []
            for ik in range(nprobe):
                l = int(ci[qi, ik])
                if l < 0 or sizes[l] == 0:
                    continue
                if metric == B.METRIC_L2:
                    tab = np.empty(tsz, np.float32)
                    R.ref_fvec_madd(tsz, B._fp(np.ascontiguousarray(table[l].reshape(-1))), -2.0, B._fp(st2), B._fp(tab))
                    dis = np.full(sizes[l], cd[qi, ik], np.float32)
                else:
                    tab = st2
                    dis = np.full(sizes[l], R.ref_fvec_inner_product(B._fp(xq), B._fp(np.ascontiguousarray(cc[l])), d),
                                  np.float32)
                tab = tab.reshape(M, 256)
                for m in range(M):                      # dis += tab[m][code[m]], m ascending, float32
                    dis = (dis + tab[m][codes[l][:, m]]).astype(np.float32)
                sv.append(dis); si.append(ids[l])
            sv, si = np.ascontiguousarray(np.concatenate(sv)), np.ascontiguousarray(np.concatenate(si))
            n = len(sv)
            hv, hi = np.empty(Rk, np.float32), np.empty(Rk, np.int64)
            so, sio = np.empty(Rk, np.float32), np.empty(Rk, np.int64)
            R.ref_heap_stream(ks, Rk, n, B._fp(sv), B._ip(si), B._fp(hv), B._ip(hi), B._fp(so), B._ip(sio))
            # the rebuilt stream IS what faiss scanned: its own search returns the same sorted table
            assert so.tobytes() == Dr[qi].tobytes() and np.array_equal(sio, Ir[qi]), (tag, qi)
            kth = so[Rk - 1]
            ntie_cut += int((sv == kth).sum() > (so == kth).sum())
            # has_rank: exact distances in ARRAY order through the k-heap
            live = hi >= 0
            ex = np.array([(R.ref_fvec_L2sqr if ks else R.ref_fvec_inner_product)(
                B._fp(xq), B._fp(np.ascontiguousarray(base[i])), d) for i in hi[live]], dtype=np.float32)
            pv, pi = np.empty(k, np.float32), np.empty(k, np.int64)
            R.ref_heap_pop_push_stream(ks, k, len(ex), B._fp(ex), B._ip(np.ascontiguousarray(hi[live])), B._fp(pv),
                                       B._ip(pi))
            fin[True][0][qi], fin[True][1][qi] = pv, pi
            # without rank: heap_reorder'ed table, first k (score window wide open)
            fin[False][0][qi], fin[False][1][qi] = so[:k], sio[:k]
        out["D_rank_" + tag], out["I_rank_" + tag] = fin[True]
        out["D_norank_" + tag], out["I_norank_" + tag] = fin[False]
        out["ncut_" + tag] = np.array([ntie_cut])
        print(name, tag, "queries with a tie at the recall_num cut:", ntie_cut, "of", nq)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def gen_blas_coarse(R, name="ivfpq_blas_c3shape", N=200000, d=128, nlist=1024, M=16, nq=2048, nprobe=32, Rk=200, k=10):
    """What "the faiss-CPU path" returns for a BATCH at the C3 shape, with the library's DEFAULT
    distance_compute_blas_threshold (20): the coarse quantizer then runs exhaustive_L2sqr_blas -- norms + MKL sgemm_
    (faiss:utils/distances.cpp:215-296,303-305) -- whose summation order no restatement can reproduce bit for bit.
    The fixture pins how far the device's default path (the k-ascending fma chain of the fp32 MFMA, = the oracle's
    mode 1) is from it: the library's coarse assignment (idx + distances), the final labels / distances of Gamma's
    search on that assignment (search_preassigned + compute_dis with has_rank: the pinned oracle run on the LIBRARY's
    assignment, cross-checked against the library's own search_preassigned table), and for comparison the same with the
    BLAS switch disabled (bit-exact territory).  Trained by the real library; base and queries come from the portable
    generator and are not stored; the lists are rebuilt by the Add path under test and checked against the stored
    sizes and checksums."""
    base = synth.sift_like(N, d=d, seed=1234)
    q = synth.sift_like(nq, d=d, seed=4321)
    r = B.RefIVFPQ(d, nlist, M, 8, B.METRIC_L2)
    R.ref_set_blas_threshold(20)                 # the library default: training, Add and search as a deployment runs them
    r.train(base[:nlist * 64])
    r.add(base)
    assert r.use_precomputed_table() == 1
    cc, pq = r.coarse_centroids(), r.pq_centroids()
    sizes = np.zeros(nlist, np.int64)
    idsum = np.zeros(nlist, np.int64)
    codesum = np.zeros(nlist, np.int64)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
    o.set_trained(cc, pq, None)
    for l in range(nlist):
        i, c = r.get_list(l)
        sizes[l] = len(i)
        idsum[l] = int(i.sum())
        codesum[l] = int((c.astype(np.int64) * (1 + np.arange(M))).sum())
        if len(i):
            assert o.add_keys(l, i, c)
    o.set_raw(base)
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    out = dict(N=N, d=d, nlist=nlist, M=M, nq=nq, nprobe=nprobe, R=Rk, k=k, cc=cc, pq=pq, list_sizes=sizes.astype(np.int32),
               list_idsum=idsum, list_codesum=codesum, base_sum=np.array([base.astype(np.float64).sum()]),
               q_sum=np.array([q.astype(np.float64).sum()]))
    for tag, thr in (("blas", 20), ("exact", 1 << 30)):
        R.ref_set_blas_threshold(thr)
        cd, ci = r.coarse(q, nprobe)
        D, I, st = o.search(q, k, nprobe, recall_num=Rk, has_rank=True, metric=B.METRIC_L2, ctx=ctx, want_stages=True,
                            preassigned=(cd, ci))
        # the library's own scan on the same assignment: its sorted top-R table holds the same (distance, id) pairs
        Dp, Ip = r.search_preassigned(q, Rk, ci, cd)
        from tests.parity import _stage_rows
        a_d, a_i = _stage_rows(Dp, Ip)
        b_d, b_i = _stage_rows(st["recall_dis"], st["recall_ids"])
        same = (a_d == b_d).all(axis=1) & (a_i == b_i).all(axis=1)
        print(name, tag, "recall-stage tables equal to the library's search_preassigned:", int(same.sum()), "of", nq,
              "(the rest: ties at the recall_num cut, where faiss's own scanner and Gamma's differ)")
        assert (a_d == b_d).all()
        out["coarse_dis_" + tag] = cd
        out["coarse_idx_" + tag] = ci.astype(np.int16)
        out["D_" + tag] = D
        out["I_" + tag] = I.astype(np.int32)
    R.ref_set_blas_threshold(1 << 30)
    # how far the canonical GEMM form (oracle mode 1 = the device's default path for >= 20 queries) is from the library
    Dg, Ig, sg = o.search(q, k, nprobe, recall_num=Rk, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=1,
                          want_stages=True)
    cib = out["coarse_idx_blas"].astype(np.int64)
    set_diff = np.array([set(a.tolist()) != set(b.tolist()) for a, b in zip(cib, sg["coarse_idx"])])
    order_diff = (cib != sg["coarse_idx"]).any(axis=1)
    lab_diff = (out["I_blas"].astype(np.int64) != Ig).any(axis=1)
    dis_bits = (out["coarse_dis_blas"].view(np.uint32) != sg["coarse_dis"].view(np.uint32)).any(axis=1)
    print(name, "canonical GEMM form vs MKL: probe sets differ in", int(set_diff.sum()), "rows, probe order in",
          int(order_diff.sum()), ", coarse distance bits in", int(dis_bits.sum()), ", final labels in", int(lab_diff.sum()),
          "of", nq, "; label mismatches outside rows whose probe set differs:", int((lab_diff & ~set_diff).sum()))
    out["n_label_diff_canonical"] = np.array([int(lab_diff.sum())])
    out["n_set_diff_canonical"] = np.array([int(set_diff.sum())])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def gen_realtime():
    """Drive the reference's real RTInvertIndex through a scripted sequence of AddKeys /
    Update / Delete / CompactIfNeed and record the observable state after each phase."""
    import ctypes as C
    RT = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libgamma_rt_ref.so"))
    RT.ref_rt_new.restype = C.c_void_p
    RT.ref_rt_new.argtypes = [C.c_int] * 5
    RT.ref_rt_add_keys.argtypes = [C.c_void_p, C.c_int, C.c_int, B._i64p, B._u8p]
    RT.ref_rt_update.argtypes = [C.c_void_p, C.c_int, C.c_int, B._u8p]
    RT.ref_rt_delete.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    RT.ref_rt_bitmap_set.argtypes = [C.c_void_p, C.c_int]
    RT.ref_rt_compact_if_need.argtypes = [C.c_void_p]
    RT.ref_rt_list_size.restype = C.c_int64
    RT.ref_rt_list_size.argtypes = [C.c_void_p, C.c_int]
    RT.ref_rt_list_capacity.restype = C.c_int64
    RT.ref_rt_list_capacity.argtypes = [C.c_void_p, C.c_int]
    RT.ref_rt_get_list.argtypes = [C.c_void_p, C.c_int, B._i64p, B._u8p]
    RT.ref_rt_vid_pos.restype = C.c_int64
    RT.ref_rt_vid_pos.argtypes = [C.c_void_p, C.c_int64]
    nlist, cs, binit, bmax, nbits = 8, 4, 16, 4096, 8192
    h = RT.ref_rt_new(nlist, cs, binit, bmax, nbits)
    rng = np.random.default_rng(5)
    ops = []      # (op, a, b, payload) script the test replays
    snaps = []
    next_vid = 0
    live = []

    def snapshot():
        st = {}
        for l in range(nlist):
            n = RT.ref_rt_list_size(h, l)
            ids = np.empty(n, np.int64)
            codes = np.empty((n, cs), np.uint8)
            if n:
                RT.ref_rt_get_list(h, l, B._ip(ids), B._up(codes))
            st[l] = (ids, codes, RT.ref_rt_list_capacity(h, l))
        vp = np.array([RT.ref_rt_vid_pos(h, v) for v in range(next_vid)], dtype=np.int64)
        return st, vp

    script = []
    for phase in range(6):
        # adds of varying size (forces several extensions)
        for _ in range(12):
            l = int(rng.integers(0, nlist))
            n = int(rng.integers(1, 40))
            keys = np.arange(next_vid, next_vid + n, dtype=np.int64)
            codes = rng.integers(0, 256, size=(n, cs)).astype(np.uint8)
            ok = RT.ref_rt_add_keys(h, l, n, B._ip(keys), B._up(codes))
            script.append(("add", l, n, codes.copy(), ok))
            if ok:
                next_vid += n
                live.extend(keys.tolist())
        # updates: some stay in list, some move
        for _ in range(10):
            vid = int(rng.choice(live))
            l = int(rng.integers(0, nlist))
            code = rng.integers(0, 256, size=cs).astype(np.uint8)
            RT.ref_rt_update(h, l, vid, B._up(code))
            script.append(("update", l, vid, code.copy(), 1))
        # deletes: set bitmap bit then Delete (search/gamma_engine.cc:810-812 order)
        dels = rng.choice(live, size=min(len(live), 25), replace=False).astype(np.int32)
        for v in dels:
            RT.ref_rt_bitmap_set(h, int(v))
        arr = (C.c_int * len(dels))(*[int(v) for v in dels])
        RT.ref_rt_delete(h, arr, len(dels))
        script.append(("delete", 0, 0, dels.astype(np.int64), 1))
        RT.ref_rt_compact_if_need(h)
        script.append(("compact", 0, 0, np.zeros(0, np.uint8), 1))
        snaps.append(snapshot())
    out = dict(nlist=nlist, cs=cs, binit=binit, bmax=bmax, nbits=nbits, nops=len(script),
               nsnaps=len(snaps))
    for i, (op, a, b, payload, ok) in enumerate(script):
        out["op_%d" % i] = np.array([{"add": 0, "update": 1, "delete": 2, "compact": 3}[op], a, b, ok])
        out["pl_%d" % i] = payload
    snap_after = [i for i, s in enumerate(script) if s[0] == "compact"]
    out["snap_after"] = np.array(snap_after)
    for si, (st, vp) in enumerate(snaps):
        out["vp_%d" % si] = vp
        for l in range(nlist):
            out["ids_%d_%d" % (si, l)] = st[l][0]
            out["codes_%d_%d" % (si, l)] = st[l][1]
        out["caps_%d" % si] = np.array([st[l][2] for l in range(nlist)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "realtime.npz"), **out)


def gen_iwpq(R):
    """A small index trained and filled by real faiss, and the bytes faiss::write_index produces
    for it: the "IwPQ" layout GammaIVFPQIndex::Dump mirrors (index/gamma_index_io.cc:16-192)."""
    import ctypes as C
    import tempfile
    d, nlist, M, N = 16, 8, 4, 700
    base = synth.sift_like(N, d=d, seed=1234)
    r = B.RefIVFPQ(d, nlist, M, 8, B.METRIC_L2)
    r.train(base)
    r.add(base)
    R.ref_ivfpq_set_nprobe.argtypes = [C.c_void_p, C.c_int]
    R.ref_ivfpq_write_index.argtypes = [C.c_void_p, C.c_char_p]
    R.ref_ivfpq_set_nprobe(r.h, 5)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ivfpq.index")
        R.ref_ivfpq_write_index(r.h, path.encode())
        blob = np.frombuffer(open(path, "rb").read(), dtype=np.uint8).copy()
    sizes, ids, codes = [], [], []
    for l in range(nlist):
        i, c = r.get_list(l)
        sizes.append(len(i))
        ids.append(i)
        codes.append(c)
    np.savez_compressed(os.path.join(OUT, "iwpq_small.npz"), d=d, nlist=nlist, M=M, N=N, nprobe=5,
                        cc=r.coarse_centroids(), pq=r.pq_centroids(),
                        list_sizes=np.array(sizes, dtype=np.int64), list_ids=np.concatenate(ids),
                        list_codes=np.concatenate(codes), file_bytes=blob)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "iwpq":     # add one fixture, leave the others alone
        gen_iwpq(B.ref())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "reservoir":  # tied key streams through the library's ReservoirTopN
        gen_reservoir(B.ref())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "blas":      # the library's default BLAS coarse path at the C3 shape
        gen_blas_coarse(B.ref())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mode0":     # table mode 0 (no precomputed table): the library's limit lowered
        R = B.ref()
        R.ref_set_blas_threshold(1 << 30)
        gen_mode0(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ties":
        R = B.ref()
        R.ref_set_blas_threshold(1 << 30)
        gen_ivfpq_ties(R, "ivfpq_ties_d32", 32, 16, 8, 1500, 6000, 48, 6, 60, 10)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ties_c4":   # the C4 shape's cuts: > 4096 lists, 64 probes, recall_num 100
        R = B.ref()
        R.ref_set_blas_threshold(1 << 30)
        gen_ivfpq_ties(R, "ivfpq_ties_c4shape", 32, 4160, 8, 12000, 24000, 32, 64, 100, 10)
        return
    if not B.have_ref():
        raise SystemExit("oracle/_ref/libgamma_ref.so missing: run make -f oracle/Makefile.ref")
    os.makedirs(OUT, exist_ok=True)
    R = B.ref()
    R.ref_set_blas_threshold(1 << 30)
    gen_prims(R)
    gen_heap(R)
    gen_reservoir(R)
    gen_ivfpq(R, "ivfpq_l2_d32", 32, 32, 8, 6000, 40, B.METRIC_L2, 6, 64)        # dsub 4
    gen_ivfpq(R, "ivfpq_l2_d64", 64, 32, 8, 6000, 40, B.METRIC_L2, 8, 100)       # dsub 8
    gen_ivfpq(R, "ivfpq_ip_d48", 48, 16, 4, 4000, 30, B.METRIC_IP, 4, 50, True)  # dsub 12
    gen_mode0(R)
    gen_ivfpq_ties(R, "ivfpq_ties_d32", 32, 16, 8, 1500, 6000, 48, 6, 60, 10)
    gen_ivfpq_ties(R, "ivfpq_ties_c4shape", 32, 4160, 8, 12000, 24000, 32, 64, 100, 10)
    gen_realtime()
    gen_blas_coarse(R)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
