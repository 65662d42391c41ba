"""GPU parity tests (run with -m gpu on an MI355X): libgamma_hip.so through its C ABI vs the
CPU oracle on the same seeded inputs.  Distances must be bit-identical; ids identical up to
the order inside exact ties (tests/parity.py)."""
import numpy as np
import pytest

from oracle import binding as B
from tests import fixtures
from tests.parity import compare_exact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case_l2():
    return fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)


@pytest.fixture(scope="module")
def hip_l2(case_l2):
    g = fixtures.load_hip(case_l2)
    yield g
    g.close()


def test_precomputed_table_bit_exact(case_l2, hip_l2):
    t_dev = hip_l2.ivfpq_table()
    t_ref = case_l2["oracle"].table()
    assert t_dev.tobytes() == t_ref.tobytes()


def test_lists_roundtrip(case_l2, hip_l2):
    o = case_l2["oracle"]
    for l in range(case_l2["nlist"]):
        ids, codes = o.get_list(l)
        gi, gc = hip_l2.get_list(l)
        assert np.array_equal(ids, gi) and np.array_equal(codes, gc)


@pytest.mark.parametrize("metric", [B.METRIC_L2, B.METRIC_IP])
@pytest.mark.parametrize("has_rank", [True, False])
@pytest.mark.parametrize("coarse_mode", [0, 1])
def test_search_parity(case_l2, hip_l2, metric, has_rank, coarse_mode):
    from gamma_amd import api
    o, q = case_l2["oracle"], case_l2["q"]
    k, nprobe, R = 10, 8, 100
    ctx = B.make_ctx(min_score=-1e30, max_score=1e30)
    D, I, st = o.search(q, k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx,
                        coarse_mode=coarse_mode, want_stages=True)
    args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank,
                          min_score=-1e30, max_score=1e30, coarse_mode=coarse_mode)
    Dg, Ig = hip_l2.ivfpq_search(q, k, args)
    sg = hip_l2.last_stages(len(q), nprobe, R)
    assert sg["coarse_dis"].tobytes() == st["coarse_dis"].tobytes()
    assert np.array_equal(sg["coarse_idx"], st["coarse_idx"])
    # recall stage: oracle has FLT_MAX neutral for empty, device has +-inf sentinel
    rd_o = st["recall_dis"].copy()
    rd_g = sg["recall_dis"].copy()
    rd_o[st["recall_ids"] == -1] = 0
    rd_g[sg["recall_ids"] == -1] = 0
    compare_exact(rd_o, st["recall_ids"], rd_g, sg["recall_ids"])
    compare_exact(D, I, Dg, Ig)
