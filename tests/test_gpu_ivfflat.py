"""GPU tests of the IVFFLAT model (include/gamma_hip.h gamma_hip_ivfflat_*; reference
index/impl/gamma_index_ivfflat.{h,cc}) through the C ABI against the oracle's restatement of
GammaIndexIVFFlat::Search: distances bit-identical at every rank, ids up to the order inside exact ties."""
import numpy as np
import pytest

from gamma_amd import api, synth
from oracle import binding as B
from tests import fixtures
from tests.parity import compare_exact

pytestmark = pytest.mark.gpu
WIDE = dict(min_score=-3e38, max_score=3e38)


def _load(case, metric):
    """a handle holding exactly the oracle's lists (ids) and raw store"""
    g = api.GammaHip(0)
    g.ivfflat_init(case["d"], case["nlist"], metric, 1000)
    g.ivfflat_set_trained(case["cc"])
    o = case["oracle"]
    lists, counts, vids = [], [], []
    for l in range(case["nlist"]):
        ids, _ = o.get_list(l)
        if len(ids):
            lists.append(l)
            counts.append(len(ids))
            vids.append(ids)
    vids = np.concatenate(vids)
    g.add_keys_batch(lists, counts, vids, np.zeros((len(vids), 1), np.uint8))
    g.raw_init(case["d"])
    g.raw_append(case["base"])
    return g


@pytest.mark.parametrize("metric,d", [(B.METRIC_L2, 32), (B.METRIC_IP, 32), (B.METRIC_L2, 100), (B.METRIC_L2, 128),
                                      (B.METRIC_IP, 64), (B.METRIC_L2, 16), (B.METRIC_IP, 96)])
def test_ivfflat_search_matches_oracle(metric, d):
    # d = 100 has no list-major variant: the (query, probe) pair kernel; the others run list-major from 16 queries x 8 probes
    case = fixtures.trained_case(d=d, nlist=64, M=d // 4, N=20000, nq=64, metric=B.METRIC_L2)
    o = case["oracle"]
    g = _load(case, metric)
    big = synth.sift_like(300, d=d, seed=99)
    try:
        for nq in (1, 7, 33, 300):
            q = case["q"][:nq] if nq <= 64 else big
            for P, k in ((1, 10), (8, 10), (8, 1), (64, 100), (3, 2000)):
                D, I = B.ivfflat_search(o, q, k, P, metric, B.make_ctx(**WIDE))
                Dg, Ig = g.ivfflat_search(q, k, api.SearchArgs(metric=metric, nprobe=P, **WIDE))
                compare_exact(D, I, Dg, Ig)
        # delete bitmap + range filter + score window
        rng = np.random.default_rng(3)
        N = case["N"]
        dead = rng.choice(N, N // 7, replace=False)
        bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        g.bitmap_upload(bm, N)
        docs = rng.choice(N, N // 2, replace=False)
        q = case["q"][:20]
        Dw, _ = B.ivfflat_search(o, q, 10, 8, metric, B.make_ctx(**WIDE))
        lo, hi = float(np.min(Dw[:, 2])), float(np.max(Dw[:, 7]))
        lo, hi = min(lo, hi), max(lo, hi)
        for kw in (WIDE, dict(min_score=lo, max_score=hi)):
            ctx = B.make_ctx(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)], **kw)
            D, I = B.ivfflat_search(o, q, 10, 8, metric, ctx)
            Dg, Ig = g.ivfflat_search(q, 10, api.SearchArgs(metric=metric, nprobe=8,
                                                            range_filters=[api.make_range_filter(docs)], **kw))
            compare_exact(D, I, Dg, Ig)
            assert not np.isin(Ig, dead).any()
    finally:
        g.close()


def test_ivfflat_add_update_delete_follow_the_reference_lists():
    """GammaIndexIVFFlat::Add / Update (gamma_index_ivfflat.cc:305-374): quantizer->assign + AddKeys / Update.  The
    device assigns (GEMM form for batches of 20 and more, as faiss), appends to the HBM lists and moves entries on
    Update; lists and searches must equal an oracle fed the same calls."""
    d, nlist, N = 32, 64, 12000
    case = fixtures.trained_case(d=d, nlist=nlist, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    base = case["base"][:N]
    o = B.OracleIVFPQ(d, nlist, 8, 8, B.METRIC_L2)
    o.set_trained(case["cc"], case["pq"], None)
    g = api.GammaHip(0)
    g.ivfflat_init(d, nlist, api.METRIC_L2, 1000)
    g.ivfflat_set_trained(case["cc"])
    g.raw_init(d)
    try:
        B.lib().go_set_assign_mode(-1)
        for i0 in range(0, N, 1000):
            xb = base[i0:i0 + 1000]
            lno = B.ivfflat_assign(o, xb)
            order = np.argsort(lno, kind="stable")
            for l in np.unique(lno):
                sel = order[lno[order] == l]
                o.add_keys(int(l), i0 + sel, np.zeros((len(sel), 8), np.uint8))
            g.raw_append(xb)
            g.add(xb, i0)
        raw = base.copy()
        o.set_raw(raw)
        for l in range(nlist):
            assert np.array_equal(o.get_list(l)[0], g.get_list(l)[0]), l
        # Update: new vector -> new list; then compaction rules are the lists' own (tested elsewhere)
        rng = np.random.default_rng(1)
        for vid in rng.choice(N, 200, replace=False):
            x = base[int(rng.integers(0, N))] + rng.integers(-2, 3, d).astype(np.float32)
            raw[vid] = x
            lno = int(B.ivfflat_assign(o, x[None])[0])
            o.update_code(lno, int(vid), np.zeros(8, np.uint8))
            lg, code = g.encode(x[None])
            assert int(lg[0]) == lno
            g.update(lno, int(vid), code[0])
            g.raw_write(int(vid), x[None])
        for l in range(nlist):
            assert np.array_equal(o.get_list(l)[0], g.get_list(l)[0]), l
        q = case["q"][:40]
        D, I = B.ivfflat_search(o, q, 10, 8, B.METRIC_L2, B.make_ctx(**WIDE))
        Dg, Ig = g.ivfflat_search(q, 10, api.SearchArgs(metric=api.METRIC_L2, nprobe=8, **WIDE))
        compare_exact(D, I, Dg, Ig)
    finally:
        B.lib().go_set_assign_mode(0)
        g.close()


@pytest.mark.parametrize("metric,d,nlist", [(B.METRIC_L2, 32, 256), (B.METRIC_IP, 100, 256), (B.METRIC_L2, 128, 1024)])
def test_ivfflat_small_batch_chain_is_the_regular_chain(metric, d, nlist):
    """Small calls (fewer (query, probe) pairs than 2 nlist) run as four launches (gamma_hip_search.cpp ivfflat_small): results
    byte for byte those of the regular chain -- deletes, a range filter, a score window, k beyond the candidates, the
    two-level selection forced -- and the oracle's."""
    case = fixtures.trained_case(d=d, nlist=nlist, M=d // 4, N=20000, nq=64, metric=B.METRIC_L2)
    o = case["oracle"]
    g = _load(case, metric)
    rng = np.random.default_rng(d + nlist)
    N = case["N"]
    try:
        for step in range(2):
            kw_f = {}
            ctx_kw = {}
            if step == 1:
                dead = rng.choice(N, N // 7, replace=False)
                bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
                np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
                g.bitmap_upload(bm, N)
                docs = rng.choice(N, N // 2, replace=False)
                kw_f = dict(range_filters=[api.make_range_filter(docs)])
                ctx_kw = dict(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)])
            for nq in (1, 3, 16, 17, 40, 64):
                q = case["q"][:nq]
                for P, k in ((1, 10), (8, 10), (4, 300), (min(64, 2 * nlist // nq - 1), 50)):
                    if nq * P >= 2 * nlist:
                        continue
                    wins = [WIDE]
                    g.set_small_path(0)
                    Dw, _ = g.ivfflat_search(q, k, api.SearchArgs(metric=metric, nprobe=P, **WIDE, **kw_f))
                    fin = Dw[np.isfinite(Dw) & (np.abs(Dw) < 1e37)]
                    if len(fin) > 4:
                        wins.append(dict(min_score=float(np.quantile(fin, 0.2)), max_score=float(np.quantile(fin, 0.8))))
                    for kw in wins:
                        args = api.SearchArgs(metric=metric, nprobe=P, **kw, **kw_f)
                        g.set_small_path(0)
                        D0, I0 = g.ivfflat_search(q, k, args)
                        for mode in (1, 3):
                            g.set_small_path(mode)
                            D1, I1 = g.ivfflat_search(q, k, args)
                            assert D0.tobytes() == D1.tobytes() and np.array_equal(I0, I1), (step, nq, P, k, mode)
                    D, I = B.ivfflat_search(o, q, k, P, metric, B.make_ctx(**WIDE, **ctx_kw))
                    g.set_small_path(1)
                    Dg, Ig = g.ivfflat_search(q, k, api.SearchArgs(metric=metric, nprobe=P, **WIDE, **kw_f))
                    compare_exact(D, I, Dg, Ig)
    finally:
        g.close()
