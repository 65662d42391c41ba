"""GPU: several GPUs behind one index object in ONE process (gamma_hip_group_*, csrc/gamma_hip_group.cpp) and the
plugins' "devices" key.  The test box has one GPU: the group's members all live on device 0 (the exchange degenerates to
device-to-device copies, everything else is the code that runs with one member per GPU).  Expected: exactly what a
single handle holding every list returns -- same lists, same distances, same ids up to the order inside ties."""
import numpy as np
import pytest

from gamma_amd import api, synth
from oracle import binding as B
from tests import fixtures
from tests.parity import compare_exact

pytestmark = pytest.mark.gpu
WIDE = dict(min_score=-3e38, max_score=3e38)


@pytest.fixture(scope="module")
def case():
    return fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)


def _single(case):
    g = api.GammaHip(0)
    g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, api.METRIC_L2, 1000)
    g.ivfpq_set_trained(case["cc"], case["pq"], None)
    g.raw_init(case["d"])
    g.raw_append(case["base"])
    for i0 in range(0, len(case["base"]), 5000):
        g.add(case["base"][i0:i0 + 5000], i0)
    return g


def _group(case, W, weights=None):
    grp = api.GammaHipGroup([0] * W)
    for m in grp.members:
        m.ivfpq_init(case["d"], case["nlist"], case["M"], 8, api.METRIC_L2, 1000)
        m.ivfpq_set_trained(case["cc"], case["pq"], None)
        m.raw_init(case["d"])
        m.raw_append(case["base"])
    grp.set_owners(weights)
    for i0 in range(0, len(case["base"]), 5000):
        grp.add(case["base"][i0:i0 + 5000], i0)
    return grp


def _same_lists(case, grp, full):
    for l in range(case["nlist"]):
        ia, ca = grp.get_list(l, case["M"])
        ib, cb = full.get_list(l)
        assert np.array_equal(ia, ib) and np.array_equal(ca, cb), l


@pytest.mark.parametrize("W,weights", [(2, "sizes"), (3, None), (4, "sizes")])
def test_group_is_the_single_handle(case, W, weights):
    full = _single(case)
    sizes = np.array([full.list_size(l) for l in range(case["nlist"])], dtype=np.int64)
    grp = _group(case, W, sizes if weights == "sizes" else None)
    try:
        owners = np.array([grp.owner(l) for l in range(case["nlist"])])
        assert set(owners.tolist()) == set(range(W))
        if weights == "sizes":     # balanced by list size: no member holds much more than its share
            load = np.array([sizes[owners == i].sum() for i in range(W)])
            assert load.max() <= 1.15 * sizes.sum() / W
        _same_lists(case, grp, full)
        for nq in (1, 7, 64, 700):
            q = synth.sift_like(nq, d=case["d"], seed=77 + nq)
            for metric, has_rank, P, R, k in ((api.METRIC_L2, True, 8, 100, 10), (api.METRIC_IP, True, 16, 64, 5),
                                              (api.METRIC_L2, False, 12, 50, 10), (api.METRIC_L2, True, 64, 200, 20)):
                a = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=has_rank, **WIDE)
                D, I = full.ivfpq_search(q, k, a)
                Dg, Ig = grp.ivfpq_search(q, k, a)
                compare_exact(D, I, Dg, Ig)
        # a request's range filter reaches every member
        allowed = np.nonzero(np.random.default_rng(5).random(len(case["base"])) < 0.3)[0]
        q = synth.sift_like(40, d=case["d"], seed=3)
        a = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=100, has_rank=True,
                           range_filters=[api.make_range_filter(allowed)], **WIDE)
        D, I = full.ivfpq_search(q, 10, a)
        Dg, Ig = grp.ivfpq_search(q, 10, a)
        compare_exact(D, I, Dg, Ig)
        assert np.isin(Ig[Ig >= 0], allowed).all()
    finally:
        grp.close()
        full.close()


def test_group_update_and_delete_route_to_the_owners(case):
    """Update across members (the list a vector leaves and the list it joins on different GPUs), a vid named twice in
    one batch, vids never added; Delete + compaction.  Reference: the single handle fed the same calls."""
    full = _single(case)
    grp = _group(case, 3)
    try:
        rng = np.random.default_rng(11)
        base = case["base"]
        N = len(base)
        vids = rng.choice(N, size=300, replace=False).astype(np.int64)
        vids = np.concatenate([vids, vids[:5], np.array([N + 50, N + 51], dtype=np.int64)])   # repeats, unknown vids
        vecs = base[rng.integers(0, N, size=len(vids))].copy()
        full.update_batch(vids, vecs)
        grp.update(vids, vecs)
        _same_lists(case, grp, full)
        b2 = base.copy()
        for v, x in zip(vids, vecs):
            if v < N:
                b2[v] = x
        for v, x in zip(vids, vecs):
            if v < N:
                full.raw_update(int(v), x)
                for m in grp.members:
                    m.raw_update(int(v), x)
        dead = rng.choice(N, size=N // 3, replace=False).astype(np.int64)
        bm = np.zeros(N // 8 + 1, np.uint8)
        np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        full.bitmap_upload(bm, N)
        full.delete(dead)
        full.compact_if_need()
        grp.each(lambda m: m.bitmap_upload(bm, N))
        grp.delete(dead)
        grp.compact_if_need()
        _same_lists(case, grp, full)
        q = synth.sift_like(200, d=case["d"], seed=9)
        a = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=100, has_rank=True, **WIDE)
        D, I = full.ivfpq_search(q, 10, a)
        Dg, Ig = grp.ivfpq_search(q, 10, a)
        compare_exact(D, I, Dg, Ig)
        assert not np.isin(Ig, dead).any()
    finally:
        grp.close()
        full.close()


def test_batched_update_is_the_sequence_of_single_updates(case):
    """gamma_hip_ivfpq_update_batch == encode(1) + Update per vid in order (what GammaIVFPQIndex::Update does)."""
    a, b = _single(case), _single(case)
    try:
        rng = np.random.default_rng(4)
        base = case["base"]
        N = len(base)
        vids = rng.choice(N, size=200, replace=False).astype(np.int64)
        vids = np.concatenate([vids, vids[:7], np.array([N + 3], dtype=np.int64)])
        vecs = base[rng.integers(0, N, size=len(vids))].copy()
        a.update_batch(vids, vecs)
        for v, x in zip(vids, vecs):
            lno, code = b.encode(x[None])
            b.update(int(lno[0]), int(v), code[0])
        for l in range(case["nlist"]):
            ia, ca = a.get_list(l)
            ib, cb = b.get_list(l)
            assert np.array_equal(ia, ib) and np.array_equal(ca, cb), l
    finally:
        a.close()
        b.close()


def test_plugin_devices_key(case):
    """HIPIVFPQ with "devices": "0,0,0" (three members on the one GPU of the test box) == the one-GPU plugin through
    Indexing (owners from the training set), Add, batched Update, Delete, Search, Dump and Load."""
    import tempfile
    from gamma_amd import plugin
    base, q = case["base"], case["q"]
    model = '{"ncentroids": %d, "nsubvector": %d, "nprobe": 8, "metric_type": "L2"%s}'
    ms = [plugin.PluginModel("HIPIVFPQ", case["d"], model % (case["nlist"], case["M"], extra), indexing_size=5000)
          for extra in ("", ', "devices": "0,0,0"')]
    try:
        for m in ms:
            m.store(base)
            assert m.indexing() == 0          # device k-means: the same seeds, the same centroids on both
            for i0 in range(0, len(base), 5000):
                assert m.add(base[i0:i0 + 5000])
        req = '{"metric_type": "L2", "recall_num": 100, "nprobe": 8}'
        for n in (len(q), 5):
            D, I = ms[0].search(q[:n], 10, req)
            Dg, Ig = ms[1].search(q[:n], 10, req)
            compare_exact(D, I, Dg, Ig)
        rng = np.random.default_rng(2)
        vids = rng.choice(len(base), size=64, replace=False).astype(np.int64)
        vecs = base[rng.integers(0, len(base), size=64)].copy()
        dead = rng.choice(len(base), size=2000, replace=False).astype(np.int64)
        b2 = base.copy()
        b2[vids] = vecs
        for m in ms:
            assert m.update_batch(vids, vecs) == 0
            assert m.delete(dead) == 0
        D, I = ms[0].search(q, 10, req)
        Dg, Ig = ms[1].search(q, 10, req)
        compare_exact(D, I, Dg, Ig)
        assert not np.isin(Ig, dead).any()
        # Dump from the sharded model, Load into a one-GPU model and the other way round
        with tempfile.TemporaryDirectory() as td:
            assert ms[1].dump(td) == 0
            m2 = plugin.PluginModel("HIPIVFPQ", case["d"], model % (case["nlist"], case["M"], ""), indexing_size=5000)
            m3 = plugin.PluginModel("HIPIVFPQ", case["d"], model % (case["nlist"], case["M"], ', "devices": "0,0"'),
                                    indexing_size=5000)
            try:
                for m in (m2, m3):
                    m.store(b2)
                    m.engine_bitmap_set(dead)
                    assert m.load(td) > 0
                    Dl, Il = m.search(q, 10, req)
                    compare_exact(Dg, Ig, Dl, Il)
            finally:
                m2.close()
                m3.close()
    finally:
        for m in ms:
            m.close()


def test_replicated_group_is_the_single_handle_bit_for_bit(case):
    """gamma_hip_group_set_placement(1): every member holds every list, a search splits the queries -- labels and
    distances are those of ONE handle strictly (compare_exact: no tie tolerance), before and after batched Update and
    Delete; the plugin's `"placement": "replicate"` key is the same thing behind the RetrievalModel boundary."""
    from gamma_amd import plugin
    full = _single(case)
    grp = api.GammaHipGroup([0] * 3)
    try:
        grp.set_placement(True)
        for m in grp.members:
            m.ivfpq_init(case["d"], case["nlist"], case["M"], 8, api.METRIC_L2, 1000)
            m.ivfpq_set_trained(case["cc"], case["pq"], None)
            m.raw_init(case["d"])
            m.raw_append(case["base"])
        grp.set_owners(None)
        for i0 in range(0, len(case["base"]), 5000):
            grp.add(case["base"][i0:i0 + 5000], i0)
        _same_lists(case, grp, full)
        for m in grp.members:
            for l in (0, 17, case["nlist"] - 1):
                assert m.list_size(l) == full.list_size(l)

        def check():
            for nq in (1, 2, 7, 64, 700):
                q = synth.sift_like(nq, d=case["d"], seed=177 + nq)
                for metric, has_rank, P, R, k in ((api.METRIC_L2, True, 8, 100, 10), (api.METRIC_IP, True, 16, 64, 5),
                                                  (api.METRIC_L2, False, 12, 50, 10)):
                    a = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=has_rank, **WIDE)
                    D, I = full.ivfpq_search(q, k, a)
                    Dg, Ig = grp.ivfpq_search(q, k, a)
                    compare_exact(D, I, Dg, Ig)
        check()
        rng = np.random.default_rng(12)
        vids = rng.choice(len(case["base"]), size=300, replace=False).astype(np.int64)
        vecs = case["base"][rng.integers(0, len(case["base"]), size=300)].copy()
        dead = rng.choice(len(case["base"]), size=1500, replace=False).astype(np.int64)
        full.update_batch(vids, vecs)
        grp.update(vids, vecs)
        full.delete(dead)
        grp.delete(dead)
        b2 = case["base"].copy()
        b2[vids] = vecs
        for g in [full] + grp.members:
            for v, x in zip(vids, vecs):
                g.raw_update(int(v), x)
        _same_lists(case, grp, full)
        check()
    finally:
        grp.close()
        full.close()
    model = '{"ncentroids": %d, "nsubvector": %d, "nprobe": 8, "metric_type": "L2"%s}'
    ms = [plugin.PluginModel("HIPIVFPQ", case["d"], model % (case["nlist"], case["M"], extra), indexing_size=5000)
          for extra in ("", ', "devices": "0,0,0", "placement": "replicate"')]
    try:
        for m in ms:
            m.store(case["base"])
            assert m.indexing() == 0
            for i0 in range(0, len(case["base"]), 5000):
                assert m.add(case["base"][i0:i0 + 5000])
        req = '{"metric_type": "L2", "recall_num": 100, "nprobe": 8}'
        for n in (len(case["q"]), 5, 1):
            D, I = ms[0].search(case["q"][:n], 10, req)
            Dg, Ig = ms[1].search(case["q"][:n], 10, req)
            compare_exact(D, I, Dg, Ig)
    finally:
        for m in ms:
            m.close()
    # Parse rejects an unknown placement
    with pytest.raises(Exception):
        plugin.PluginModel("HIPIVFPQ", case["d"], model % (case["nlist"], case["M"], ', "devices": "0,0", "placement": "x"'),
                           indexing_size=5000)


@pytest.mark.parametrize("tag,W", [("l2", 2), ("ip", 3), ("l2", 4)])
def test_sharded_group_keeps_the_reference_order_inside_ties(tag, W):
    """List-sharded members and tie-heavy data (every base vector four times): the owner of a query slice lists the
    queries a tie can change, every member exports their candidate streams over the lists it owns, the owner replays
    them -- labels strictly the pinned oracle's on the unsharded index."""
    from tests.test_oracle_golden import load_ties
    z, o, base, metric = load_ties(tag)
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    sizes = z["list_sizes_" + tag]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    grp = api.GammaHipGroup([0] * W)
    try:
        for m in grp.members:
            m.ivfpq_init(d, nlist, M, 8, metric)
            m.ivfpq_set_trained(z["cc_" + tag], z["pq_" + tag], None)
            m.raw_init(d)
            m.raw_append(base)
        grp.set_owners(sizes)
        for l in range(nlist):
            if sizes[l]:
                grp.add_keys(l, z["list_ids_" + tag][offs[l]:offs[l + 1]], z["list_codes_" + tag][offs[l]:offs[l + 1]])
        ctx = B.make_ctx(**WIDE)
        for nprobe, R, k, has_rank in ((12, 60, 10, True), (6, 40, 10, False), (16, 100, 20, True)):
            D1, I1 = o.search(z["q"], k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx, coarse_mode=0)
            a = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **WIDE)
            for reps in (1, 15):
                q = np.tile(z["q"], (reps, 1))[:len(z["q"]) * reps - (reps > 1)]
                Dg, Ig = grp.ivfpq_search(q, k, a)
                compare_exact(np.tile(D1, (reps, 1))[:len(q)], np.tile(I1, (reps, 1))[:len(q)], Dg, Ig)
            # with the mode off for the request only the distances agree on this data
            a0 = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, exact_ties=-1, **WIDE)
            Dg, Ig = grp.ivfpq_search(z["q"], k, a0)
            assert Dg.tobytes() == D1.tobytes() or not has_rank
    finally:
        grp.close()


def test_group_exchanges_through_rccl_where_a_communicator_forms(case):
    """gamma_hip_group_set_transport(g, 1): the assignment goes through ONE in-place ncclAllGather and the per-shard tables
    through one grouped ncclSend / ncclRecv exchange (the north star's RCCL over xGMI), one communicator per member.  The test
    box has one GPU: a group of ONE member forms a (one-rank) communicator and runs every RCCL call of the path; members that
    share a device cannot, stay on copies and say so.  Results are the single handle's either way."""
    full = _single(case)
    q = case["q"]
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=60, has_rank=True, **WIDE)
    D0, I0 = full.ivfpq_search(q, 10, args)
    one = _group(case, 1)
    two = _group(case, 2)
    try:
        one.set_transport(True)
        D1, I1 = one.ivfpq_search(q, 10, args)
        t1 = one.transport()
        assert t1["rccl"] and t1["rccl_searches"] == 1 and t1["note"] == "RCCL", t1
        compare_exact(D0, I0, D1, I1)
        D1b, I1b = one.ivfpq_search(q[:7], 10, args)        # another batch size through the same communicator
        compare_exact(D0[:7], I0[:7], D1b, I1b)
        assert one.transport()["rccl_searches"] == 2
        two.set_transport(True)
        D2, I2 = two.ivfpq_search(q, 10, args)
        t2 = two.transport()
        assert not t2["rccl"] and t2["rccl_searches"] == 0 and "share a device" in t2["note"], t2
        compare_exact(D0, I0, D2, I2)
    finally:
        one.close()
        two.close()
        full.close()
