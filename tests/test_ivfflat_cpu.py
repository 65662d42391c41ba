"""CPU: the oracle's IVFFLAT restatement (oracle/gamma_oracle.c go_ivfflat_search; reference
index/impl/gamma_index_ivfflat.{h,cc}) against what pins it: probing EVERY list is the flat search, whose loop and
primitives are pinned against compiled faiss (tests/test_oracle_golden.py, test_oracle_vs_ref.py)."""
import numpy as np
import pytest

from oracle import binding as B
from tests import fixtures
from tests.parity import compare_topk


@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
def test_ivfflat_over_all_lists_is_the_flat_search(metric):
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    o, q = case["oracle"], case["q"][:24]
    N = case["N"]
    rng = np.random.default_rng(2)
    dead = rng.choice(N, N // 9, replace=False)
    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
    np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
    for ctx_kw in (dict(), dict(docids_bitmap=bm)):
        D, I = B.ivfflat_search(o, q, 10, case["nlist"], metric, B.make_ctx(**ctx_kw))
        Df, If = B.flat_search(case["base"], q, 10, metric, B.make_ctx(**ctx_kw))
        compare_topk(Df, If, D, I)


def test_ivfflat_probe_subset_and_stages():
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    o, q = case["oracle"], case["q"][:16]
    D, I, st = B.ivfflat_search(o, q, 5, 4, B.METRIC_L2, B.make_ctx(), want_stages=True)
    Dc, Ic = B.knn_L2sqr(q, case["cc"], 4, mode=0)          # quantizer->search below 20 queries: the exact form
    assert st["coarse_dis"].tobytes() == Dc.tobytes() and np.array_equal(st["coarse_idx"], Ic)
    # every result comes from a probed list and is the exact distance
    for qi in range(len(q)):
        members = np.concatenate([o.get_list(int(l))[0] for l in Ic[qi]]) & 0x7fffffffffffffff
        ok = I[qi] >= 0
        assert np.isin(I[qi][ok], members).all()
        ex = ((case["base"][I[qi][ok]] - q[qi]) ** 2).sum(axis=1)
        assert np.allclose(D[qi][ok], ex, rtol=1e-5)
        assert np.all(np.diff(D[qi][ok]) >= 0)
